"""Two-pass shooting against the one-pass kernel (GUSTO_SHOOT_ONE_PASS=1): same results, timing.  python tools/shoot_ab.py [B]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import gusto_jl_amd as g
P = g.problems
B = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
s = g.BatchSolver(g.DUBINS_CAR, 30, B, hist_cap=64)
s.set_problems(*P.dubins_batch(B)); s.solve(3)
out = {}
for mode in ("two", "one"):
    if mode == "one": os.environ["GUSTO_SHOOT_ONE_PASS"] = "1"
    else: os.environ.pop("GUSTO_SHOOT_ONE_PASS", None)
    ts = []
    for _ in range(2):
        t0 = time.perf_counter(); r = s.shoot(); ts.append(time.perf_counter() - t0)
    out[mode] = r
    print(mode, "pass: %.1f ms" % (1e3 * min(ts)), "optimal", int((r["status"] == 1).sum()), "mean iters %.2f max %d" % (r["newton_iters"].mean(), r["newton_iters"].max()))
a, b = out["two"], out["one"]
ok = a["status"] == 1
print("status equal", np.array_equal(a["status"], b["status"]), "iters equal", np.array_equal(a["newton_iters"], b["newton_iters"]),
      "resid equal", np.array_equal(a["resid"], b["resid"], equal_nan=True), "p0 equal", np.array_equal(a["p0"], b["p0"], equal_nan=True),
      "X equal (converged)", np.array_equal(a["X"][ok], b["X"][ok]), "U equal", np.array_equal(a["U"][ok], b["U"][ok]))
