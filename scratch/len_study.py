import sys, os, time
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/oracle')
import numpy as np
import gusto_jl_amd as g
import gusto_oracle as go
P = g.problems
env = P.freeflyer_env()
B = 4096
x0, glo, ghi, tf = P.freeflyer_batch(B)
t0 = time.time()
r = go.solve_batch(go.FREEFLYER_SE2, 50, env, None, x0, glo, ghi, tf, 30, 8)
print('oracle', time.time() - t0, 's; converged', r['converged'].sum())
np.savez('scratch/len_study.npz', x0=x0, its=r['iterations'], ipm=r['ipm_iters'], conv=r['converged'])
print('its', np.bincount(r['iterations']))
ipm = r['ipm_iters']
print('ipm total', ipm.sum(), 'max', ipm.max(), 'mean', ipm.mean())
print('quantiles', np.percentile(ipm, [50, 90, 95, 99, 99.5, 99.9, 100]))
