"""One, two and four waves per problem (gusto_set_decomposition; csrc/segw.hpp) of the 12/13-state models over batch sizes, with the
shipped library: python tools/chains_sweep.py [out.jsonl] -- one JSON line per (model, batch): HIP-event kernel time of a whole
gusto_solve (best of 3) per decomposition, what AUTO takes, converged problems and KKT solves."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import gusto_jl_amd as g
P = g.problems
out = open(sys.argv[1], "w") if len(sys.argv) > 1 else None
boxes, spheres = P.iss_corner_env(True)
for model, name, gen, sizes in ((g.ASTROBEE_SE3, "astrobeeSE3", P.astrobee_se3_batch, (128, 256, 512, 768, 1024, 1536, 2048, 4096, 8192)),
                                (g.ASTROBEE_SE3_MANIFOLD, "astrobeeSE3manifold", P.astrobee_manifold_batch, (128, 256, 512, 768, 1024, 1536, 2048, 4096))):
    for B in sizes:
        batch = gen(B)
        e = {"model": name, "N": 50, "B": B}
        for key, dec in (("one_wave", 1), ("two_waves", 3), ("four_waves", 4), ("auto", 0)):
            s = g.BatchSolver(model, 50, B, hist_cap=64, boxes=boxes, spheres=spheres)
            s.set_decomposition(dec)
            ms = []
            for rep in range(4):
                s.set_problems(*batch); s.solve(30)
                if rep: ms.append(s.last_solve_ms())
            st = s.status()
            e[key + "_ms"] = round(min(ms), 3)
            e[key + "_kkt"] = int(st["ipm_iters"].sum())
            e[key + "_scp"] = int(st["iterations"].sum())
            e[key + "_converged"] = int(st["converged"].sum())
            if dec == 0:
                e["auto_lds_bytes"] = s.launch_info()[1]
            s.close()
        e["auto_is"] = min(("one_wave", "two_waves", "four_waves"), key=lambda k: abs(e[k + "_ms"] - e["auto_ms"]))
        print(json.dumps(e), flush=True)
        if out: out.write(json.dumps(e) + "\n"); out.flush()
