"""One launch of a bench workload plus a calibration copy of known size, for rocprofv3 --pmc runs.
   python tools/pmc_probe.py [config 2..5] [gusto|trajopt]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import gusto_jl_amd as g
import bench
P = g.problems
cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 2
algo = sys.argv[2] if len(sys.argv) > 2 else "gusto"
c = bench.CONFIGS[cfg]
B = c["B"] if algo == "gusto" else {2: 1024, 4: 256}[cfg]
model, boxes, spheres, (x0, glo, ghi, tf) = bench.workload(P, g, cfg, B, 0)
if algo == "gusto":
    s = g.BatchSolver(model, c["N"], B, hist_cap=64, boxes=boxes, spheres=spheres)
    s.set_problems(x0, glo, ghi, tf)
    s.solve(30)
else:
    s = g.TrajOptSolver(model, c["N"], B, boxes=boxes, spheres=spheres)
    s.set_problems(x0, glo, ghi, tf)
    s.solve(125)
st = s.status()
print("kernel_ms", s.last_solve_ms(), "ipm", int(st["ipm_iters"].sum()), "scp", int(st["iterations"].sum()), "config", cfg, "algo", algo, "B", B)
# calibration: 1 GiB read + 1 GiB write through a float64 elementwise kernel (8 B per lane, like the solver's accesses)
a = torch.zeros(1 << 27, dtype=torch.float64, device="cuda")
torch.cuda.synchronize()
b = a + 1.0
torch.cuda.synchronize()
