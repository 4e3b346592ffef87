"""dubins_car: the lane-per-problem kernel against the wave-per-problem kernel on the same batch (statuses, trip counts,
interior point iterations, trajectories) and their kernel times.  usage: python tools/lane_check.py [B] [reps]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gusto_jl_amd as g  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
N = 30
x0, glo, ghi, tf = g.problems.dubins_batch(B)
out = {}
for name, dec in (("wave", 1), ("lane", 2)):
    s = g.BatchSolver(g.DUBINS_CAR, N, B, hist_cap=40)
    s.set_decomposition(dec)
    ms = []
    for _ in range(reps):
        s.set_problems(x0, glo, ghi, tf)
        s.solve(30)
        ms.append(s.last_solve_ms())
    X, U = s.traj()
    st = s.status()
    out[name] = (X, U, st, ms)
    print(f"{name}: kernel ms {['%.2f' % m for m in ms]}  converged {int(st['converged'].sum())}  "
          f"total_ipm {int(st['ipm_iters'].sum())}  trips {int(st['iterations'].sum())}  slots {s.launch_info()}", flush=True)
    del s
Xw, Uw, sw, _ = out["wave"]
Xl, Ul, sl, _ = out["lane"]
same_it = sw["iterations"] == sl["iterations"]
same_stop = sw["stop_reason"] == sl["stop_reason"]
print(f"same trips {same_it.mean():.5f}  same stop {same_stop.mean():.5f}  same ipm {(sw['ipm_iters'] == sl['ipm_iters']).mean():.5f}")
ok = same_it & same_stop & sw["converged"]
d = np.abs(Xw[ok] - Xl[ok]).reshape(ok.sum(), -1).max(1)
print(f"converged & same path: {ok.sum()}  |dX| median {np.median(d):.2e}  99% {np.quantile(d, 0.99):.2e}  max {d.max():.2e}")
# loop iterations of a lane kernel wave = the largest (interior point iterations + trips) among its 64 problems
w = (sl["ipm_iters"] + sl["iterations"] + (sl["stop_reason"] == 2))
per_wave = w[: (B // 64) * 64].reshape(-1, 64).max(1) if B >= 64 else w
print(f"work per problem: mean {w.mean():.1f} max {w.max()}  per wave (max of 64): mean {per_wave.mean():.1f} min {per_wave.min()} max {per_wave.max()}")
print("quantiles of per-problem work:", np.quantile(w, [0.5, 0.75, 0.9, 0.99, 0.999]))
