"""How long does the longest problem of the config-2 batch take when it has the GPU (nearly) to itself?
python tools/longest_alone.py [B] -- solves the batch, picks the problems with the most KKT solves, re-solves them alone / in small groups."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import gusto_jl_amd as g
P = g.problems
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
x0, glo, ghi, tf = P.freeflyer_batch(B); boxes = P.freeflyer_env()
s = g.BatchSolver(0, 50, B, hist_cap=64, boxes=boxes)
for rep in range(2):
    s.set_problems(x0, glo, ghi, tf); s.solve(30)
st = s.status()
ipm = st["ipm_iters"]; o = np.argsort(-ipm)
print(f"batch B={B}: kernel {s.last_solve_ms():.2f} ms; KKT solves total {ipm.sum()} mean {ipm.mean():.1f} max {ipm.max()}; top problems {o[:8].tolist()} with {ipm[o[:8]].tolist()} KKT solves, SCP iterations {st['iterations'][o[:8]].tolist()}")
for cnt in (1, 4, 64, 256, 1024):
    idx = o[:cnt]
    s2 = g.BatchSolver(0, 50, cnt, hist_cap=64, boxes=boxes)
    for rep in range(2):
        s2.set_problems(x0[idx], glo[idx], ghi[idx], tf[idx] if np.ndim(tf) else tf); s2.solve(30)
    i2 = s2.status()["ipm_iters"]
    print(f"  the {cnt} longest alone: kernel {s2.last_solve_ms():.2f} ms, max KKT {i2.max()}, -> {1e3 * s2.last_solve_ms() / i2.max():.1f} us per KKT solve of the longest problem")
