"""developer probe: 3-D emissive shell of 5120 triangles (sampling sweeps, shadow rays) at the origin and far from it, and the guided 3-D solve on it,
against the oracle"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "..")))
import bench
from elaina_amd import UniformIntegratorSettings
from elaina_amd.guided import GuidedIntegratorSettings
from elaina_amd.integrator3d import GuidedIntegrator3, Problem3, UniformIntegrator3, default_net_config3
from oracle.oracle import Oracle, default_net_config3 as ocfg3, guided_settings3
orc = Oracle()
for centre in ((0.0, 0.0, 0.0), (300.0, -150.0, 200.0)):
    c = np.asarray(centre, np.float32)
    V, T = bench.icosphere(4, 1.0)
    Vi, Ti = bench.icosphere(2, 0.45)
    V = (V + c).astype(np.float32); Vi = (Vi + c).astype(np.float32)
    ncol = np.repeat((0.3 * (V[:, 1:2] - c[1])).astype(np.float32), 6, axis=1)
    sd = {"d_verts": Vi, "d_tris": Ti, "d_colors": np.repeat((Vi[:, :1] - c[0]), 6, axis=1).astype(np.float32), "n_verts": V, "n_tris": T,
          "n_colors": ncol, "probe": (0.7, tuple(float(x) for x in c), (0.0, 1.0, 0.0), (1.0, 0.0, 0.0)), "dirichlet_intensity": 1.0, "neumann_intensity": 1.0}
    w, h, spp, depth, eps = 20, 20, 3, 48, 2e-3
    it = UniformIntegrator3(Problem3.from_dict(sd), UniformIntegratorSettings((w, h), spp, depth, eps))
    it.solve()
    ref = orc.solve3(sd, w, h, spp, depth, eps, threads=os.cpu_count() or 8)
    print(centre, "emissive 5120: steps", it.last_stats["walk_steps"], ref["walk_steps"], "field equal", np.array_equal(it.solution.reshape(-1, 3), ref["field"]),
          "max diff", float(np.abs(it.solution.reshape(-1, 3) - ref["field"]).max()), flush=True)
    it.close()
    aabb = (tuple(float(x) for x in c - 1.1), tuple(float(x) for x in c + 1.1))
    st = GuidedIntegratorSettings(frameSize=(w, h), samplesPerPixel=4, trainSppCount=2, maxWalkingDepth=32, epsilonShell=eps, batchSize=512, minBatchSize=64)
    gi = GuidedIntegrator3(Problem3.from_dict(sd), st, aabb, network_config=default_net_config3(n_levels=4), seed=7)
    p0 = gi.network.params()
    gi.solve()
    gs = guided_settings3(w, h, 4, 32, eps, aabb[0], aabb[1], train_spp_count=2, batch_size=512, min_batch_size=64)
    trained = p0.copy()
    gref = orc.solve_guided3(sd, gs, ocfg3(n_levels=4), trained, threads=16, dump_spp=-1)
    print(centre, "guided: steps", gi.last_stats["walk_steps"], gref["walk_steps"], "field equal", np.array_equal(gi.solution, gref["field"]),
          "params equal", np.array_equal(gi.network.params(), trained), flush=True)
    gi.close()
