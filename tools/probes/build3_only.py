"""the device mesh build alone (configs.mesh_build3 of bench.py): the command the build3 stage of tools/gpu_round.sh traces"""
import json, os, sys
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "..")))
import bench
class Env: local = 0
print(json.dumps(bench.run_mesh_build3(Env)), flush=True)
