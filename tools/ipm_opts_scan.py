"""Sensitivity of a config to the interior point options (runtime knobs of gusto_ipm_opts): python tools/ipm_opts_scan.py <model> <B> <N>
prints, per setting: kernel ms, converged problems, total interior point iterations, trips."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import gusto_jl_amd as g
P = g.problems
model, B, N = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
FIRST = int(os.environ.get("FIRST", "0"))   # offset into the config's problem generator (another batch of the same distribution)
boxes = spheres = None
if model == 0:
    x0, glo, ghi, tf = P.freeflyer_batch(B, FIRST); boxes = P.freeflyer_env()
elif model == 1:
    x0, glo, ghi, tf = P.dubins_batch(B, FIRST)
elif model == 2:
    x0, glo, ghi, tf = P.astrobee_se3_batch(B, FIRST); boxes, spheres = P.iss_corner_env(True)
else:
    x0, glo, ghi, tf = P.astrobee_manifold_batch(B, FIRST); boxes, spheres = P.iss_corner_env(True)
# MUW = list of  floor[:gain[:max]]  settings; "auto" = the model's defaults (common.hpp: warm_defaults)
floors = [float(x) for x in os.environ.get("MUFLOOR", "0").split(",")]   # 0 = the default complementarity floor
accs = [int(x) for x in os.environ.get("ACC", "-1").split(",")]           # -1 = the default acc_iter
for spec, fl, ac in [(a, b, c) for a in os.environ.get("MUW", "auto,1e-4").split(",") for b in floors for c in accs]:
    io = g.default_ipm_opts()
    if fl > 0: io.mu_floor = fl
    if ac >= 0: io.acc_iter = ac
    if os.environ.get("SIGMAX"): io.sigma_max = float(os.environ["SIGMAX"])
    mw = spec
    if spec != "auto":
        f = [float(x) for x in spec.split(":")]
        io.mu_warm = f[0]; io.mu_warm_gain = f[1] if len(f) > 1 else 0.0; io.mu_warm_max = f[2] if len(f) > 2 else f[0]
    s = g.BatchSolver(model, N, B, hist_cap=64, boxes=boxes, spheres=spheres, ipm_opts=io)
    for rep in range(2):
        s.set_problems(x0, glo, ghi, tf); s.solve(30)
    st = s.status()
    print(f"model {model} mu_warm {mw} mu_floor {io.mu_floor:g} acc_iter {io.acc_iter} sigma_max {io.sigma_max:g}: kernel {s.last_solve_ms():.1f} ms conv {st['converged'].sum()} succ {st['successful'].sum()} "
          f"ipm {st['ipm_iters'].sum()} trips {st['iterations'].sum()} stops {np.bincount(st['stop_reason'], minlength=5)}", flush=True)
    del s
