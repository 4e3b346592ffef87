from nb import *
import ctypes as C, time
from concurrent.futures import ThreadPoolExecutor
x0,glo,ghi,tf=P.freeflyer_batch(4096)
env=P.freeflyer_env()
def mk(trtol):
    io=go.IpmOpts(tol=1e-8,tol_acc=1e-5,mu_floor=1e-11,tr_tol=trtol,mu_warm=1e-4,max_iter=60); return io
def solve(args):
    b,trtol=args
    o=go.Oracle(go.FREEFLYER_SE2,50,boxes=env,ipm_opts=mk(trtol))
    o.set_problem(x0[b],glo[b],ghi[b],tf[b]); r=o.solve(30)
    return r
t=time.time()
with ThreadPoolExecutor(8) as ex:
    A=list(ex.map(solve,[(b,1e-6) for b in range(4096)]))
    Bz=list(ex.map(solve,[(b,0.0) for b in range(4096)]))
print('time',time.time()-t)
ndec=nflip=nprob=0; ntrv_a=ntrv_b=0
for a,b in zip(A,Bz):
    n=min(len(a['tr_sat']),len(b['tr_sat']))
    # first decision where they differ (afterwards the runs are different problems)
    d=np.nonzero(a['tr_sat'][:n]!=b['tr_sat'][:n])[0]
    ndec+=len(a['tr_sat'])-1; ntrv_a+=int((a['tr_sat'][1:]==0).sum()); ntrv_b+=int((b['tr_sat'][1:]==0).sum())
    if len(d): nprob+=1
print('decisions',ndec,'problems whose trace departs',nprob,'TR-violated verdicts: tol 1e-6:',ntrv_a,' tol 0:',ntrv_b)
print('converged a',sum(r['converged'] for r in A),'b',sum(r['converged'] for r in Bz))
print('iters a',sum(r['iterations'] for r in A),'b',sum(r['iterations'] for r in Bz))
