import sys, time, numpy as np
sys.path.insert(0, '/root/repo')
from elaina_amd import Problem
from oracle.oracle import Oracle
o = Oracle()
p = Problem.load_scene("ladybug")
sd = p.as_dict()
depth, eps = p.default_max_depth, p.default_eps
frame = 1024
rows = list(range(4, 1024, 16))
spps=[1,2,4,8,16,24,32,48,64,96,128,160,192,224]
cum={s:[] for s in spps}
t=time.time()
for r in rows:
    for s in spps:
        a = o.solve(sd, frame, frame, s, depth, eps, pixel_begin=r*frame, pixel_end=(r+1)*frame, threads=8, want_steps=True)
        cum[s].append(a["steps"].copy())
print(time.time()-t)
np.savez('/tmp/wost_sim/cum.npz', spps=np.array(spps), **{"c%d"%s: np.array(cum[s]) for s in spps})
