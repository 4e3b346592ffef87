"""Bitwise regression check of a kernel change: python tools/gpu_bits.py <model> <B> save|check
   save: writes devdata/bits_m<model>_<B>.npz (X, U, iterations, ipm_iters of one solve); check: compares with it."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import gusto_jl_amd as g
P = g.problems
model, B, mode = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
N = 30 if model == 1 else 50
boxes = spheres = None
if model == 0:
    batch = P.freeflyer_batch(B); boxes = P.freeflyer_env()
elif model == 1:
    batch = P.dubins_batch(B)
elif model == 2:
    batch = P.astrobee_se3_batch(B); boxes, spheres = P.iss_corner_env(True)
else:
    batch = P.astrobee_manifold_batch(B); boxes, spheres = P.iss_corner_env(True)
s = g.BatchSolver(model, N, B, hist_cap=64, boxes=boxes, spheres=spheres)
s.set_problems(*batch); s.solve(30)
X, U = s.traj(); st = s.status()
path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "devdata", f"bits_m{model}_{B}.npz")
if mode == "save":
    os.makedirs(os.path.dirname(path), exist_ok=True)
    np.savez(path, X=X, U=U, it=st["iterations"], ipm=st["ipm_iters"])
    print("saved", path, "ipm total", st["ipm_iters"].sum())
else:
    d = np.load(path)
    same = np.array_equal(d["X"], X) and np.array_equal(d["U"], U) and np.array_equal(d["it"], st["iterations"]) and np.array_equal(d["ipm"], st["ipm_iters"])
    dx = np.abs(d["X"] - X).max(); du = np.abs(d["U"] - U).max()
    print(f"model {model} B={B}: bit-identical {same}; max|dX| {dx:.3e} max|dU| {du:.3e}; iterations equal {np.array_equal(d['it'], st['iterations'])} "
          f"({(d['it'] != st['iterations']).sum()} differ); ipm total {st['ipm_iters'].sum()} vs {d['ipm'].sum()}; kernel {s.last_solve_ms():.1f} ms")
