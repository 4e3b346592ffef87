"""Prototype check (CPU): full GuSTO solves by the oracle with its sequential Riccati KKT solve against the SEGMENTED one
(tools/proto/segriccati.c, built by tools/proto/build.sh).  python tools/proto/compare.py MODEL B [S] -- runs itself twice as
child processes (the oracle library is chosen at import time), then compares per problem: SCP iterations, converged flags,
interior point iterations, final trajectories."""
import os, sys, subprocess, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np


def run(model, B, out, first):
    import gusto_jl_amd as g
    import gusto_oracle as go
    P = g.problems
    env = sph = None
    N = 50
    if model == 0:
        env = P.freeflyer_env(); x0, glo, ghi, tf = P.freeflyer_batch(B, first=first)
    elif model == 1:
        N = 30; x0, glo, ghi, tf = P.dubins_batch(B, first=first)
    elif model == 2:
        env, sph = P.iss_corner_env(True); x0, glo, ghi, tf = P.astrobee_se3_batch(B, first=first)
    else:
        env, sph = P.iss_corner_env(True); x0, glo, ghi, tf = P.astrobee_manifold_batch(B, first=first)
    N = int(os.environ.get("CMP_N", N))
    r = go.solve_batch(model, N, env, sph, x0, glo, ghi, tf, 30, 0)
    np.savez(out, **{k: np.asarray(v) for k, v in r.items()})


if __name__ == "__main__":
    if sys.argv[1] == "child":
        run(int(sys.argv[2]), int(sys.argv[3]), sys.argv[4], int(sys.argv[5]))
        sys.exit(0)
    model, B = int(sys.argv[1]), int(sys.argv[2])
    S = sys.argv[3] if len(sys.argv) > 3 else "2"
    first = sys.argv[4] if len(sys.argv) > 4 else "0"
    d = tempfile.mkdtemp()
    res = []
    for tag, lib in (("seq", None), ("seg", os.path.join(ROOT, "scratch/proto/libgusto_oracle_seg.so"))):
        e = dict(os.environ)
        if lib: e["GUSTO_ORACLE_LIB"] = lib; e["GO_SEG_S"] = S
        o = os.path.join(d, tag + ".npz")
        subprocess.check_call([sys.executable, __file__, "child", str(model), str(B), o, first], env=e)
        res.append(np.load(o))
    a, b = res
    same_it = a["iterations"] == b["iterations"]
    same_cv = a["converged"] == b["converged"]
    dx = np.abs(a["X"] - b["X"]).reshape(B, -1).max(1)
    print(f"model {model} B={B} S={S} first={first} NU={os.environ.get('GO_SEG_NU','1')} FWD={os.environ.get('GO_SEG_FWD','0')}: identical SCP iterations {same_it.mean()*100:.2f}%  converged flags {same_cv.mean()*100:.2f}%")
    print(f"  KKT solves seq {a['ipm_iters'].sum()} seg {b['ipm_iters'].sum()}  ({(b['ipm_iters'].sum()/a['ipm_iters'].sum()-1)*100:+.3f}%)  converged seq {a['converged'].sum()} seg {b['converged'].sum()}")
    both = same_it & a["converged"].astype(bool) & b["converged"].astype(bool)
    if both.any():
        print(f"  converged, same iterations ({both.sum()}): max|dX| median {np.median(dx[both]):.2e} 99% {np.quantile(dx[both], .99):.2e} max {dx[both].max():.2e}")
    di = np.abs(a["ipm_iters"].astype(int) - b["ipm_iters"].astype(int))
    print(f"  problems with different KKT-solve counts: {(di > 0).sum()}  max diff {di.max()}; different SCP iterations: {[(int(i), int(a['iterations'][i]), int(b['iterations'][i])) for i in np.where(~same_it)[0][:10]]}")
    if os.environ.get("CMP_WORST"):
        d = b["ipm_iters"].astype(int) - a["ipm_iters"].astype(int)
        o = np.argsort(-np.abs(d))[:12]
        print("  worst:", [(int(i), int(a["ipm_iters"][i]), int(b["ipm_iters"][i])) for i in o])
