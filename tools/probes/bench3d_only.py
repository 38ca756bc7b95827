"""the 3-D entries of bench.py alone: one pass (or argv[1] passes) of run_uniform3d -- the command the pmc3d stage of tools/gpu_round.sh profiles"""
import os, sys, json
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "..")))
import bench
class Env: local = 0
class Args: no_cpu_baseline = True
for k in range(int(sys.argv[1]) if len(sys.argv) > 1 else 1):
    print(json.dumps({n: (round(e["kernel_ms"], 1), "%.3g" % e["value"]) for n, e in bench.run_uniform3d(Env, Args).items()}), flush=True)
