import numpy as np, heapq
R=np.load('sched_R.npy'); L=R[:,0]; B=len(L); probe=R[:,6]; om2=R[:,5]
rng=np.random.default_rng(0)
def makespan(order, cost, slots=1024):
    h=[0.0]*slots; heapq.heapify(h)
    for b in order: t=heapq.heappop(h); heapq.heappush(h,t+cost[b])
    return max(h)
print('random orders', [makespan(rng.permutation(B),L) for _ in range(5)])
print('in-order',makespan(range(B),L))
# two-launch: probe pass then sorted pass (barrier between)
t1=makespan(range(B),probe); rem=L-probe
lvl=np.round(np.log10(om2)).astype(int)
order=np.argsort(-lvl,kind='stable')
print('two-launch: probe',t1,'+ sorted',makespan(order,rem),'=',t1+makespan(order,rem))
# dynamic single launch: event simulation
def dynamic(slots=1024, key=lvl):
    import bisect
    t_free=[(0.0,s) for s in range(slots)]; heapq.heapify(t_free)
    nextA=0; ready=[]  # heap of (-key, b) with availability time
    pending=[]  # (avail_time, -key, b)
    end=0.0; done=0
    while done<B:
        t,s=heapq.heappop(t_free)
        # move pending -> ready for avail<=t
        while pending and pending[0][0]<=t:
            a,k,b=heapq.heappop(pending); heapq.heappush(ready,(k,b))
        if nextA<B:
            b=nextA; nextA+=1
            tf=t+probe[b]
            if rem[b]>0: heapq.heappush(pending,(tf,-key[b],b))
            else: done+=1; end=max(end,tf)
            heapq.heappush(t_free,(tf,s))
        elif ready:
            k,b=heapq.heappop(ready); tf=t+rem[b]; done+=1; end=max(end,tf); heapq.heappush(t_free,(tf,s))
        elif pending:
            a=pending[0][0]; heapq.heappush(t_free,(a,s))   # wait for the next probe to finish
        else:
            break
    return end
print('dynamic (omega level key)',dynamic())
print('dynamic (oracle key = remaining)',dynamic(key=rem))
print('dynamic (key=probe ipm)',dynamic(key=probe))
# variant: run long-looking problems to completion right after their probe (no requeue) -- i.e. in-order but others deferred
