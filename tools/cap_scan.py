"""Effect of gusto_ipm_opts.max_iter on a config: python tools/cap_scan.py <config 2..5> <cap> [cap ...]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import gusto_jl_amd as g
import bench
P = g.problems
cfg = int(sys.argv[1])
c = bench.CONFIGS[cfg]
model, boxes, spheres, batch = bench.workload(P, g, cfg, c["B"], 0)
for cap in [int(v) for v in sys.argv[2:]]:
    io = g.default_ipm_opts(); io.max_iter = cap
    s = g.BatchSolver(model, c["N"], c["B"], hist_cap=64, boxes=boxes, spheres=spheres, ipm_opts=io)
    for rep in range(2):
        s.set_problems(*batch); s.solve(30)
    st = s.status()
    print(f"config {cfg} cap {cap}: kernel {s.last_solve_ms():.1f} ms conv {st['converged'].sum()} ipm {st['ipm_iters'].sum()} stops {np.bincount(st['stop_reason'], minlength=5)}")
    s.close()
