import sys
sys.path.insert(0,'/root/repo/oracle'); sys.path.insert(0,'/root/repo')
import numpy as np, gusto_oracle as go, gusto_jl_amd as g
x0,glo,ghi,tf = g.problems.dubins_batch(6)
o = go.Oracle(go.DUBINS_CAR, 30)
b=4
print(x0[b])
o.set_problem(x0[b],glo[b],ghi[b],tf[b])
X,U = o.init_straightline()
r = o.subproblem(X,U,1e4,1.0,1e4/8+0.01)
print({k:v for k,v in r.items() if k not in ('X','U','dual')})
# batch stats
x0,glo,ghi,tf = g.problems.dubins_batch(2000)
r = go.solve_batch(go.DUBINS_CAR, 30, None, None, x0, glo, ghi, tf, 30, 8)
print('conv', r['converged'].mean(), 'iter0', (r['iterations']==0).mean(), np.bincount(r['iterations'],minlength=31))
