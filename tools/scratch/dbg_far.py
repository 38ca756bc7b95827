import sys, os
sys.path.insert(0, "/root/repo")
from elaina_amd import Problem, UniformIntegrator, UniformIntegratorSettings
p = Problem.load_scene("ladybug")
for frame, spp in ((64, 8), (256, 64)):
    it = UniformIntegrator(p, UniformIntegratorSettings((frame, frame), spp, 64, 1.0))
    it.solve()
    print(frame, spp, it.last_stats["walk_steps"], it.last_stats["kernel_launches"])
    it.close()
