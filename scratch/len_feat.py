import sys
sys.path.insert(0, '/root/repo')
import numpy as np
import gusto_jl_amd as g
P = g.problems
d = np.load('scratch/len_study.npz')
x0, ipm, its = d['x0'], d['ipm'], d['its']
env = P.freeflyer_env()
goal = P.FREEFLYER_X_GOAL
B = len(x0)
t = np.linspace(0, 1, 50)[None, :, None]
pts = (1 - t) * x0[:, None, 0:2] + t * goal[None, None, 0:2]       # B,50,2
def sdf(p, lo, hi):
    dx = np.maximum(np.maximum(lo[0] - p[..., 0], 0), p[..., 0] - hi[0])
    dy = np.maximum(np.maximum(lo[1] - p[..., 1], 0), p[..., 1] - hi[1])
    out = np.hypot(dx, dy)
    ins = -np.minimum(np.minimum(p[..., 0] - lo[0], hi[0] - p[..., 0]), np.minimum(p[..., 1] - lo[1], hi[1] - p[..., 1]))
    return np.where((dx > 0) | (dy > 0), out, ins)
D = np.stack([sdf(pts, bx[0:2], bx[3:5]) for bx in env], -1) - P.FREEFLYER_RADIUS   # B,50,nobs
viol = np.maximum(0.05 - D, 0)      # clearance 0.05?
f_sum = viol.sum((1, 2)); f_max = viol.max((1, 2)); f_cnt = (viol > 0).sum((1, 2)); f_len = np.hypot(*(x0[:, 0:2] - goal[0:2]).T)
f_nobs = (viol > 0).any(1).sum(1)
for name, f in [('sum', f_sum), ('max', f_max), ('cnt', f_cnt), ('len', f_len), ('nobs', f_nobs)]:
    order = np.argsort(-f, kind='stable')
    rank = np.empty(B, int); rank[order] = np.arange(B)
    long_ = np.argsort(-ipm)[:64]
    print(name, 'corr', np.corrcoef(f, ipm)[0, 1], 'ranks of 64 longest: max', rank[long_].max(), 'frac in first 1024:', (rank[long_] < 1024).mean(),
          'longest 8 ranks', rank[long_[:8]])
# simulate makespan: list scheduling on S slots with per-unit time 1
import heapq
def makespan(order, S=1024):
    h = [0.0] * S
    heapq.heapify(h)
    for b in order:
        t0 = heapq.heappop(h); heapq.heappush(h, t0 + ipm[b])
    return max(h)
print('FCFS', makespan(np.arange(B)), 'perfect LPT', makespan(np.argsort(-ipm)), 'balanced', ipm.sum() / 1024, 'longest', ipm.max())
for name, f in [('sum', f_sum), ('max', f_max), ('cnt', f_cnt), ('nobs', f_nobs), ('sum+len', f_sum + 0.01 * f_len)]:
    print(name, makespan(np.argsort(-f, kind='stable')))
