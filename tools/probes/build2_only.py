"""the 2-D tree builds of bench.py alone (ladybug and fille, device against host, five builds each way): the command the build2 stage
of tools/gpu_round.sh traces"""
import os, sys, json
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "..")))
import bench
class Env: local = 0
print("mesh_build2 " + json.dumps(bench.run_mesh_build2(Env)), flush=True)
