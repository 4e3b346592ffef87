#!/bin/bash
# Debug-build LDS index asserts: builds every model with -DGUSTO_DEBUG_LDS (common.hpp: every indexed
# LDS access through an LPtr is checked against the workgroup's allocation and traps past it) and runs whole solves of a few
# problems per model, one-wave and multi-wave horizons, GuSTO and TrajOpt.  Runs on the GPU box: gpurun -- 'bash tools/debug_lds.sh'
# (the in-tree library is REPLACED by the debug build: rebuild with __graft_entry__.build() afterwards).
cd ${GRAFT_REPO_ROOT:-$(dirname "$0")/..}
mkdir -p gpurun_out
python - <<'PY' > gpurun_out/debug_lds.log 2>&1
import subprocess, sys, os
sys.path.insert(0, os.getcwd())
import numpy as np
flags = "-DGUSTO_DEBUG_LDS"
ok = True
for model, Ns in ((0, (50, 130)), (1, (30,)), (2, (50,)), (3, (50,)), (4, (30,)), (5, (20,)), (6, (20,))):
    r = subprocess.run(["bash", "tools/build_dev.sh", str(model), flags], capture_output=True, text=True)
    if r.returncode:
        print("build failed", model, r.stdout[-2000:], r.stderr[-2000:]); ok = False; continue
    code = f'''
import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np, gusto_jl_amd as g
P = g.problems
model = {model}
for N in {Ns}:
    pub = {{0: 0, 1: 1, 2: 2, 3: 3, 4: 0, 5: 2, 6: 3}}[model]
    bx, sp = (P.freeflyer_env(), None) if pub == 0 else ((None, None) if pub == 1 else P.iss_corner_env(True))
    batch = {{0: P.freeflyer_batch, 1: P.dubins_batch, 2: P.astrobee_se3_batch, 3: P.astrobee_manifold_batch}}[pub](6)
    if model >= 4:
        s = g.TrajOptSolver(pub, N, 6, boxes=bx, spheres=sp); s.set_problems(*batch); s.solve(6)
    else:
        s = g.BatchSolver(pub, N, 6, hist_cap=40, boxes=bx, spheres=sp); s.set_problems(*batch); s.solve(8)
        for dec in ((1, 3, 4) if model in (2, 3) else ()):   # one / two / four waves per problem (csrc/segw.hpp; the default above: four)
            d = g.BatchSolver(pub, N, 6, hist_cap=40, boxes=bx, spheres=sp); d.set_decomposition(dec); d.set_problems(*batch); d.solve(8)
            print("model", model, "N", N, "decomposition", dec, "ok", bool(np.isfinite(d.traj()[0]).all()), d.status()["iterations"])
            if not np.array_equal(d.status()["iterations"], s.status()["iterations"]):   # (a debug build once ran the four-wave helpers' sweeps into NaN)
                print("out of bounds or worse: decomposition", dec, "disagrees with the default"); sys.exit(1)
    X, U = s.traj()
    print("model", model, "N", N, "ok", bool(np.isfinite(X).all()), s.status()["iterations"])
'''
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True)
    print(r.stdout[-1500:], r.stderr[-1500:])
    if r.returncode or "out of bounds" in r.stdout + r.stderr:
        ok = False
print("LDS index asserts:", "clean" if ok else "FAILED")
PY
tail -30 gpurun_out/debug_lds.log
