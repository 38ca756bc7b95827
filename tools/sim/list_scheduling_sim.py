import numpy as np, heapq
z = np.load('/tmp/wost_sim/steps.npz')
rows, steps, s16, sdf = z['rows'], z['steps'].astype(np.int64), z['steps16'].astype(np.float64), z['sdf']
d0 = np.abs(sdf.reshape(1024,1024)[rows,:]).ravel()
st = steps.ravel(); n=len(st)
LANES = 393216//16
def makespan(order, lanes=LANES):
    # list scheduling: each lane takes next pixel in order when free
    h=[0]*lanes
    heapq.heapify(h)
    for i in order:
        t=heapq.heappop(h); heapq.heappush(h,t+st[i])
    return max(h)
ideal = st.sum()/LANES
print("ideal", ideal, "max", st.max(), "lower bound", max(ideal, st.max()))
rng=np.random.default_rng(0)
print("row order", makespan(np.arange(n))/ideal)
print("random", makespan(rng.permutation(n))/ideal)
print("exact LPT", makespan(np.argsort(-st))/ideal)
print("d0 LPT", makespan(np.argsort(-d0,kind='stable'))/ideal)
# bucketed d0 (log2 buckets, 1/4 octave)
b=np.floor(np.log2(np.maximum(d0,0.25))*4)
print("d0 bucket LPT", makespan(np.argsort(-b,kind='stable'))/ideal)
b=np.floor(np.log2(np.maximum(d0,0.25))*2)
print("d0 half-octave bucket LPT", makespan(np.argsort(-b,kind='stable'))/ideal)
b=np.floor(np.log2(np.maximum(d0,0.25)))
print("d0 octave bucket LPT", makespan(np.argsort(-b,kind='stable'))/ideal)
print("16spp LPT", makespan(np.argsort(-s16.ravel(),kind='stable'))/ideal)
# current round scheme: rounds of 256 steps, time per round = ceil(nactive/lanes)*256 (roughly)
rem=st.copy(); T=0
while (rem>0).any():
    na=(rem>0).sum(); 
    # blocks of the round are scheduled as they finish; approximate per-lane: each lane slot takes min(rem,256)
    work=np.minimum(rem[rem>0],256)
    h=[0]*LANES; heapq.heapify(h)
    for w in work:
        t=heapq.heappop(h); heapq.heappush(h,t+256)   # lane held until the block ends ~ 256 steps
    T+=max(h); rem=np.maximum(rem-256,0)
print("round scheme (block held 256)", T/ideal)
