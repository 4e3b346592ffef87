"""One launch of the bench workload plus a calibration copy of known size, for rocprofv3 --pmc runs."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import gusto_jl_amd as g
P = g.problems
B = 4096
x0, glo, ghi, tf = P.freeflyer_batch(B)
s = g.BatchSolver(g.FREEFLYER_SE2, 50, B, hist_cap=64, boxes=P.freeflyer_env())
s.set_problems(x0, glo, ghi, tf)
s.solve(30)
st = s.status()
print("kernel_ms", s.last_solve_ms(), "ipm", int(st["ipm_iters"].sum()), "scp", int(st["iterations"].sum()))
# calibration: 1 GiB read + 1 GiB write through a float64 elementwise kernel (8 B per lane, like the solver's accesses)
a = torch.zeros(1 << 27, dtype=torch.float64, device="cuda")
torch.cuda.synchronize()
b = a + 1.0
torch.cuda.synchronize()
