"""the 3-D entries of bench.py alone: one pass (or argv[1] passes) of run_uniform3d -- the command the pmc3d stage of tools/gpu_round.sh profiles"""
import os, sys, json
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "..")))
import bench
class Env: local = 0
class Args: no_cpu_baseline = True
# BENCH3D_FRAME=1024: the full-chip frames (a million walkers) instead of the 512^2 ones
big = os.environ.get("BENCH3D_FRAME") == "1024"
only = ("dirichlet_icosphere_1280_1024", "neumann_shell_1280_1024") if big else ("dirichlet_icosphere_1280", "neumann_shell_1280")
for k in range(int(sys.argv[1]) if len(sys.argv) > 1 else 1):
    print(json.dumps({n: (round(e["kernel_ms"], 1), "%.3g" % e["value"]) for n, e in bench.run_uniform3d(Env, Args, only=only).items()}), flush=True)
