from nb import *
import heapq, pickle, os
from concurrent.futures import ThreadPoolExecutor
B=4096
x0,glo,ghi,tf=P.freeflyer_batch(B); env=P.freeflyer_env()
def solve(b):
    o=go.Oracle(go.FREEFLYER_SE2,50,boxes=env); o.set_problem(x0[b],glo[b],ghi[b],tf[b]); r=o.solve(30)
    return r['ipm_iters'][1:].astype(int), r['omega'][1:]
if os.path.exists('trips.pkl'): D=pickle.load(open('trips.pkl','rb'))
else:
    with ThreadPoolExecutor(8) as ex: D=list(ex.map(solve,range(B)))
    pickle.dump(D,open('trips.pkl','wb'))
L=np.array([d[0].sum() for d in D])
print('balanced',L.sum()/1024,'longest',L.max())
def lvl(w): return int(round(np.log10(max(w,1.0))))
def sim(policy, slots=1024, slice_trips=1, ovh=0.0):
    # state per problem: next trip index
    nxt=[0]*B
    ready=[]   # heap (-prio, seq, b)
    seq=0
    for b in range(B):
        heapq.heappush(ready,(0,seq,b)); seq+=1
    free=[(0.0,s) for s in range(slots)]; heapq.heapify(free)
    pending=[] # (time available, prio, seq, b)
    end=0.0; remaining=B
    while remaining>0:
        t,s=heapq.heappop(free)
        while pending and pending[0][0]<=t:
            a,pr,sq,b=heapq.heappop(pending); heapq.heappush(ready,(pr,sq,b))
        if not ready:
            if not pending: break
            heapq.heappush(free,(pending[0][0],s)); continue
        pr,sq,b=heapq.heappop(ready)
        ip,om=D[b]
        k0=nxt[b]; k1=min(len(ip),k0+slice_trips(k0) if callable(slice_trips) else k0+slice_trips)
        cost=ip[k0:k1].sum()+ovh
        nxt[b]=k1; tfin=t+cost
        if k1>=len(ip): remaining-=1; end=max(end,tfin)
        else:
            pr2=policy(b,k1,om)
            heapq.heappush(pending,(tfin,pr2,seq,b)); seq+=1
        heapq.heappush(free,(tfin,s))
    return end
pol_lvl=lambda b,k,om: -lvl(om[k-1])
pol_lvl_trips=lambda b,k,om: -(lvl(om[k-1])*100+k)      # higher omega first, then more trips done (older) first
pol_lvl_young=lambda b,k,om: -(lvl(om[k-1])*100-k)
for name,pol in (('omega level',pol_lvl),('level then most-trips',pol_lvl_trips),('level then fewest-trips',pol_lvl_young)):
    for sl in (1,2,3):
        print(name,'slice',sl,'makespan',sim(pol,slice_trips=sl,ovh=0.5))
# probe 2 trips then run to completion by level (the dynamic single-launch scheme)
print('probe2+completion', sim(pol_lvl, slice_trips=lambda k0: 2 if k0==0 else 10**6, ovh=0.5))
print('probe1+completion', sim(pol_lvl, slice_trips=lambda k0: 1 if k0==0 else 10**6, ovh=0.5))
print('probe1, then 1 more, then completion', sim(pol_lvl, slice_trips=lambda k0: 1 if k0<2 else 10**6, ovh=0.5))
