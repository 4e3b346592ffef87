import sys
sys.path.insert(0,'/root/repo/oracle'); sys.path.insert(0,'/root/repo')
import numpy as np, gusto_oracle as go, gusto_jl_amd as g
P=g.problems; env=P.freeflyer_env(); N=50
B=4096
x0,glo,ghi,tf = P.freeflyer_batch(B)
r = go.solve_batch(go.FREEFLYER_SE2, N, env, None, x0, glo, ghi, tf, 30, 8)
ipm=r['ipm_iters']; it=r['iterations']
print('total ipm', ipm.sum(), 'mean', ipm.mean(), 'max', ipm.max())
print('percentiles', np.percentile(ipm,[50,90,99,99.9,100]))
order=np.argsort(-ipm)[:15]
for b in order: print(b, 'ipm', ipm[b], 'trips', it[b], 'conv', r['converged'][b], 'per-trip', ipm[b]/max(1,it[b]))
# work if perfectly balanced on 1024 slots vs longest
print('balanced bound (iterations per slot)', ipm.sum()/1024, 'longest', ipm.max())
