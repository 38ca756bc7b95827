import numpy as np, heapq
z = np.load('/tmp/wost_sim/steps.npz'); c=np.load('/tmp/wost_sim/cum.npz')
rows, steps, sdf = z['rows'], z['steps'].astype(np.int64), z['sdf']
d0 = np.abs(sdf.reshape(1024,1024)[rows,:]).ravel()
st = steps.ravel(); n=len(st)
spps=list(c['spps'])+[256]
cum=np.stack([np.zeros(n)]+[c['c%d'%s].ravel().astype(np.float64) for s in spps[:-1]]+[st.astype(np.float64)],1)   # n x (len+1)
sp=np.array([0]+spps,dtype=np.float64)
LANES = 393216//16
b=np.floor(np.log2(np.maximum(d0*d0,1e-6))*4)
order=np.argsort(-b,kind='stable')
h=[(0,-1)]*LANES; heapq.heapify(h)
for i in order:
    t,_=heapq.heappop(h); heapq.heappush(h,(t+st[i],i)); tl=t
surv=[(t-tl,i) for t,i in h if t>tl]
rem=np.array([r for r,i in surv],dtype=np.float64); idx=np.array([i for r,i in surv])
done_steps=st[idx]-rem
# samples done: interpolate on the cumulative curve of each pixel
sdone=np.array([np.interp(done_steps[k],cum[idx[k]],sp) for k in range(len(idx))])
sfl=np.floor(sdone)
est=(256-sfl)*np.maximum(done_steps,4)/np.maximum(sfl,1)
print("survivors",len(rem)*16,"T_dry",tl)
print("samples done pct",np.percentile(sfl,[1,5,10,25,50,75]))
err=est/np.maximum(rem,1)
for lo,hi in [(0,8),(8,16),(16,32),(32,64),(64,128),(128,256)]:
    m=(sfl>=lo)&(sfl<hi)
    if m.sum(): print("done %3d..%3d: n %6d  true rem mean %5.0f max %5.0f   est/true pct5 %.2f pct50 %.2f pct95 %.2f; underestimated by >512 steps: %d; true rem>1024: %d, of which est<1024: %d"%(lo,hi,m.sum()*16,rem[m].mean(),rem[m].max(),*np.percentile(err[m],[5,50,95]),((rem-est)[m]>512).sum()*16,(rem[m]>1024).sum()*16,((rem[m]>1024)&(est[m]<1024)).sum()*16))
def policy(name, score, theta, cap=None):
    L=score>=theta
    print("%-40s long: %6d  (true rem mean %4.0f)  max true rem among the rest: %4.0f; rest with rem>768: %5d, >1024: %5d, >1280: %5d"%(name,L.sum()*16,rem[L].mean() if L.sum() else 0,rem[~L].max(),(rem[~L]>768).sum()*16,(rem[~L]>1024).sum()*16,(rem[~L]>1280).sum()*16))
policy("est>=1024",est,1024)
policy("est>=768",est,768)
ucb=est*(1+2/np.sqrt(np.maximum(sfl,1)))
policy("ucb2>=1024",ucb,1024)
ucb=est*(1+3/np.sqrt(np.maximum(sfl,1)))
policy("ucb3>=1024",ucb,1024)
policy("ucb3>=1280",ucb,1280)
mix=np.where(sfl<16,1e9,est)
policy("done<16 or est>=1024",mix,1024)
mix=np.where(sfl<32,1e9,est)
policy("done<32 or est>=1024",mix,1024)
np.savez('/tmp/wost_sim/surv.npz',rem=rem,idx=idx,est=est,sfl=sfl,done_steps=done_steps)
