import sys, time
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/oracle')
import numpy as np
from multiprocessing import Pool
import gusto_jl_amd as g
import gusto_oracle as go
P = g.problems
env = P.freeflyer_env()
B = 1024
x0, glo, ghi, tf = P.freeflyer_batch(B)
def work(a):
    lo, hi, kw = a
    io = go.IpmOpts(tol=1e-8, tol_acc=1e-5, mu_floor=1e-11, tr_tol=1e-6, mu_warm=1e-4, max_iter=60)
    for k, v in kw.items(): setattr(io, k, v)
    o = go.Oracle(go.FREEFLYER_SE2, 50, boxes=env, ipm_opts=io)
    out = []
    for b in range(lo, hi):
        o.set_problem(x0[b], glo[b], ghi[b], tf[b])
        r = o.solve(30)
        out.append((r['total_ipm_iters'], r['iterations'], r['converged'], r['successful'], r['J_true'][-1]))
    return out
if __name__ == '__main__':
    with Pool(8) as p:
        base = None
        for kw in [dict(), dict(mu_warm=1e-3), dict(mu_warm=1e-5), dict(mu_warm=1e-6), dict(mu_warm=1e-2), dict(tol=1e-7), dict(tol=1e-6), dict(mu_floor=1e-10)]:
            res = [x for ch in p.map(work, [(i, i + 32, kw) for i in range(0, B, 32)]) for x in ch]
            a = np.array(res, float)
            if base is None: base = a
            print(kw, 'ipm', int(a[:, 0].sum()), 'trips', int(a[:, 1].sum()), 'conv', int(a[:, 2].sum()), 'succ', int(a[:, 3].sum()),
                  'ipm/trip %.2f' % (a[:, 0].sum() / a[:, 1].sum()), 'max ipm', int(a[:, 0].max()), 'same trips', int((a[:, 1] == base[:, 1]).sum()),
                  'max |dJ| rel %.2e' % np.nanmax(np.abs(a[:, 4] - base[:, 4]) / (1e-12 + np.abs(base[:, 4]))))
