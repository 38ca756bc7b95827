import sys, json
sys.path.insert(0, '.')
import bench
class Env: local = 0
class Args: no_cpu_baseline = False
print(json.dumps(bench.run_guided3d(Env, Args)))
print(json.dumps(bench.run_neumann2d(Env, Args)))
