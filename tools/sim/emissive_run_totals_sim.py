import sys, numpy as np
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
from test_gpu_3d import _shell_scene
sd=_shell_scene(2,3,flux=lambda x,y,z:0.3*y)
V=np.asarray(sd["n_verts"],np.float64); T=np.asarray(sd["n_tris"],np.int64)
dV=np.asarray(sd["d_verts"],np.float64)
print("neumann tris",len(T),"dirichlet verts",len(dV), "shell radius", np.linalg.norm(V,axis=1).mean(), "inner radius", np.linalg.norm(dV,axis=1).mean())
P0,P1,P2=V[T[:,0]],V[T[:,1]],V[T[:,2]]
C=(P0+P1+P2)/3
rin=np.linalg.norm(dV,axis=1).mean(); rout=np.linalg.norm(V,axis=1).mean()
rng=np.random.default_rng(0)
n=len(T)
old_it=[];new_it=[];acc=[]
for _ in range(400):
    d=rng.normal(size=3); d/=np.linalg.norm(d); r=rng.uniform(rin,rout); q=d*r
    R=0.99*(r-rin)        # distance to the inner Dirichlet ball
    # approximate triangle test by the centroid and a vertex test for "fully inside"
    dv=np.linalg.norm(np.stack([P0,P1,P2],0)-q,axis=2)   # 3 x n
    touch=(dv.min(0)<=R); inside=(dv.max(0)<=R)
    # iterations of the current sweep: groups of 4 that are not skipped (touching or inside runs), ignoring the skip hierarchy's own iterations
    g4_t=touch.reshape(-1,4).any(1); 
    pad=(-n)%64
    t64=np.concatenate([touch,np.zeros(pad,bool)]).reshape(-1,64); i64=np.concatenate([inside,np.ones(pad,bool)]).reshape(-1,64)
    full64=i64.all(1)
    old=g4_t.sum()
    new=full64.sum()+ (t64.any(1)&~full64).sum()*0 + np.concatenate([g4_t,np.zeros(pad//4,bool)]).reshape(-1,16)[~full64].sum()
    old_it.append(old);new_it.append(new);acc.append(touch.sum())
print("accepted triangles per query: mean %.0f max %d"%(np.mean(acc),max(acc)))
print("groups of 4 visited: now %.0f, with per-64 totals %.0f  (ratio %.2f)"%(np.mean(old_it),np.mean(new_it),np.mean(old_it)/max(np.mean(new_it),1)))
