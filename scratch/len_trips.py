import sys, pickle
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/oracle')
import numpy as np
from multiprocessing import Pool
import gusto_jl_amd as g
import gusto_oracle as go
P = g.problems
B = 4096
x0, glo, ghi, tf = P.freeflyer_batch(B)
env = P.freeflyer_env()
def work(rng):
    o = go.Oracle(go.FREEFLYER_SE2, 50, boxes=env)
    out = []
    for b in range(*rng):
        o.set_problem(x0[b], glo[b], ghi[b], tf[b])
        r = o.solve(30)
        out.append((r['ipm_iters'][1:].copy(), r['omega'].copy(), r['iterations']))
    return out
if __name__ == '__main__':
    with Pool(8) as p:
        res = p.map(work, [(i, i + 64) for i in range(0, B, 64)])
    flat = [x for ch in res for x in ch]
    pickle.dump(flat, open('scratch/trips.pkl', 'wb'))
    print(sum(x[0].sum() for x in flat))
