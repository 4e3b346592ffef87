from nb import *
import time, heapq
from concurrent.futures import ThreadPoolExecutor
from scipy.stats import spearmanr
B=4096
x0,glo,ghi,tf=P.freeflyer_batch(B); env=P.freeflyer_env()
def solve(b):
    o=go.Oracle(go.FREEFLYER_SE2,50,boxes=env); o.set_problem(x0[b],glo[b],ghi[b],tf[b]); r=o.solve(30)
    Xi,Ui=o.init_straightline()
    d=np.array([[o.signed_distance(0,Xi[k,:2],i)[0] for i in range(14)] for k in range(50)])
    ip=r['ipm_iters']
    return r['total_ipm_iters'], d.min(), (d.min(1)<0).sum(), (d.min(1)<0.05).sum(), np.linalg.norm(x0[b,:2]-glo[b,:2]), r['omega'][min(2,len(r['omega'])-1)], ip[1:3].sum(), r['iterations'], -np.minimum(d,0).sum()
with ThreadPoolExecutor(8) as ex: R=np.array(list(ex.map(solve,range(B))))
L=R[:,0]
for i,nm in enumerate(['mind','npen','nclose','dist','omega2','ipm_first2','iters','pensum'],1):
    print(nm, 'spearman %.3f'%spearmanr(R[:,i],L)[0])
def makespan(order, slots=1024):
    h=[0.0]*slots; heapq.heapify(h)
    for b in order: t=heapq.heappop(h); heapq.heappush(h,t+L[b])
    return max(h)
print('balanced',L.sum()/1024,'longest',L.max())
print('in-order',makespan(range(B)),'LPT',makespan(np.argsort(-L)))
for i,nm in enumerate(['mind','npen','nclose','dist','omega2','ipm_first2','iters','pensum'],1):
    key=R[:,i]*(-1 if nm=='mind' else 1)
    print('sorted by',nm,makespan(np.argsort(-key,kind='stable')))
np.save('sched_R.npy',R)
