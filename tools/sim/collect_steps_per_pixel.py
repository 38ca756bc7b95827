import sys, time, numpy as np
sys.path.insert(0, '/root/repo')
from elaina_amd import Problem
from oracle.oracle import Oracle
o = Oracle()
p = Problem.load_scene("ladybug")
sd = p.as_dict()
depth, eps = p.default_max_depth, p.default_eps
print(depth, eps)
frame = 1024
rows = list(range(4, 1024, 16))
steps = []; steps16=[]
t=time.time()
for r in rows:
    a = o.solve(sd, frame, frame, 256, depth, eps, pixel_begin=r*frame, pixel_end=(r+1)*frame, threads=8, want_steps=True)
    steps.append(a["steps"].copy())
    b = o.solve(sd, frame, frame, 16, depth, eps, pixel_begin=r*frame, pixel_end=(r+1)*frame, threads=8, want_steps=True)
    steps16.append(b["steps"].copy())
print(time.time()-t)
d = o.render_dirichlet_sdf(sd, frame, frame, threads=8)
np.savez('/tmp/wost_sim/steps.npz', rows=np.array(rows), steps=np.array(steps), steps16=np.array(steps16), sdf=np.asarray(d))
