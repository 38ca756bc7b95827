"""the guided 3-D entries of bench.py alone (one pass of run_guided3d without the oracle band): the command the pmc_guided3d stage of
tools/gpu_round.sh profiles"""
import os, sys, json
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "..")))
import bench
class Env: local = 0
class Args: no_cpu_baseline = True
print(json.dumps({n: (round(e["ms_per_step"], 1), "%.3g" % e["value"]) for n, e in bench.run_guided3d(Env, Args).items()}), flush=True)
