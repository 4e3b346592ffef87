"""The drop-in boundary without a GPU: every symbol of include/gusto_hip.h is exported, the library fails loudly
when no device exists (there is no CPU fallback), and the host-side mirror of the reference API behaves."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import gusto_jl_amd as g

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "gusto_hip.h")).read()
    declared = set(re.findall(r"\b(gusto_[a-z_0-9]+)\s*\(", hdr))
    declared -= {"gusto_history"}
    assert declared == set(g._capi.SYMBOLS), declared ^ set(g._capi.SYMBOLS)
    L = g.lib()
    for s in declared:
        assert hasattr(L, s), s


def test_struct_layouts_match_the_header():
    sp, mp = g.default_params(g.FREEFLYER_SE2)
    assert C.sizeof(g.ScpParams) == 10 * 8
    assert C.sizeof(g.ModelParams) == (1 + 3 + 2 + 4 + 4 + 13 + 13) * 8 + 8 + 6 * 8
    # per-model defaults of the reference (freeflyer_se2.jl:15-39, robot/freeflyer.jl:28-62)
    assert (sp.Delta0, sp.omega0, sp.omega_max, sp.eps, sp.rho0, sp.rho1) == (3.0, 1.0, 1e10, 1e-2, 0.1, 0.3)
    assert (sp.beta_succ, sp.beta_fail, sp.gamma_fail, sp.convergence_threshold) == (2.0, 0.5, 10.0, 1e-2)
    assert abs(mp.mass - 16.72) < 1e-12 and abs(mp.hard_limit_accel - 0.37 / 16.72) < 1e-15
    assert abs(mp.hard_limit_alpha - 0.593 / (0.184 / 6.43)) < 1e-12 and mp.n_robot_comp == 2
    sp, mp = g.default_params(g.DUBINS_CAR)
    assert (sp.Delta0, sp.eps, sp.rho0, sp.rho1, sp.gamma_fail, sp.convergence_threshold) == (1e4, 1e-6, 0.4, 1.5, 5.0, 1e-4)
    sp, mp = g.default_params(g.ASTROBEE_SE3_MANIFOLD)
    assert (sp.Delta0, sp.eps, sp.rho1, sp.convergence_threshold) == (1e3, 1e-1, 100.0, 1e-4) and mp.mass == 7.0
    n, m = C.c_int(), C.c_int()
    for mid, dims in g._capi.MODEL_DIMS.items():
        assert g.lib().gusto_model_dims(mid, C.byref(n), C.byref(m)) == 0 and (n.value, m.value) == dims
    assert g.lib().gusto_model_dims(7, C.byref(n), C.byref(m)) == -1


@pytest.mark.skipif(_has_gpu(), reason="checks the no-GPU failure mode")
def test_no_gpu_means_loud_failure_not_cpu_fallback():
    with pytest.raises(g.GustoError) as e:
        g.BatchSolver(g.FREEFLYER_SE2, 50, 4)
    assert "-4" in str(e.value) and "no CPU fallback" in str(e.value)


def test_bad_arguments_are_rejected():
    L = g.lib()
    h = C.c_void_p()
    assert L.gusto_create(C.byref(h), 0, 2, 4, 16, 0) == -1       # N < 3
    assert L.gusto_create(C.byref(h), 9, 50, 4, 16, 0) == -1      # unknown model
    assert L.gusto_create(None, 0, 50, 4, 16, 0) == -1
    # entry points added on top of the reference's interface reject a null handle the same way
    assert L.gusto_solve_async(None, 30, 0) == -1
    assert L.gusto_wait(None) == -1
    assert L.gusto_set_schedule(None, 2, 2048) == -1
    assert L.gusto_set_decomposition(None, 1) == -1


def test_product_never_imports_the_oracle():
    """The product path must not route through oracle/ (it is test infrastructure)."""
    pkg = os.path.join(ROOT, "gusto.jl_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hpp", ".hip", ".h", ".jl")):
                txt = open(os.path.join(dirpath, f)).read()
                assert "gusto_oracle" not in txt and "oracle/" not in txt, os.path.join(dirpath, f)


def test_host_mirror_goal_flattening_and_straight_line():
    H = g.host
    model = H.AstrobeeSE3Manifold()
    gs = H.GoalSet()
    r_goal, q_goal = np.array([10.9, 3.0, 5.0]), np.array([1.0, 0.2, 0.3, 0.4]) / np.linalg.norm([1.0, 0.2, 0.3, 0.4])
    H.add_goal(gs, H.Goal(H.PointGoal(r_goal), 10.0, range(0, 3)))
    H.add_goal(gs, H.Goal(H.PointGoal(np.zeros(3)), 10.0, range(3, 6)))
    H.add_goal(gs, H.Goal(H.BoxGoal(q_goal - 1e-4, q_goal + 1e-4), 10.0, range(6, 10)))
    H.add_goal(gs, H.Goal(H.PointGoal(np.zeros(3)), 10.0, range(10, 13)))
    lo, hi = H._goal_bounds(gs, 13, 10.0)
    assert np.array_equal(lo[:3], r_goal) and np.array_equal(hi[:3], r_goal)
    assert np.allclose(hi[6:10] - lo[6:10], 2e-4) and np.all(lo[10:] == 0) and np.all(hi[10:] == 0)
    x_init = np.concatenate([[11.2, -0.8, 5.6], np.zeros(3), [1, 0, 0, 0], np.zeros(3)])
    PD = H.ProblemDefinition(H.Robot(), model, H.ISSCorner(True), x_init, gs)
    TOP = H.TrajectoryOptimizationProblem(PD, 50, 10.0, fixed_final_time=True)
    traj = H.init_traj_straightline(TOP)
    assert traj.X.shape == (13, 50) and traj.U.shape == (6, 50) and abs(traj.dt - 10.0 / 49) < 1e-15
    assert np.allclose(traj.X[:, 0], x_init) and np.allclose(traj.X[6:10, -1], q_goal)     # centre of the box goal
    # quaternions are interpolated linearly, i.e. not unit norm mid-way (SURVEY.md a5 note)
    assert np.linalg.norm(traj.X[6:10, 25]) < 1.0
    # free final time: accepted, and as inert as in the reference (tests/test_host_cpu.py)
    assert H.TrajectoryOptimizationProblem(PD, 50, 10.0, fixed_final_time=False).tf_guess == 10.0


def test_problem_generators_are_reproducible():
    P = g.problems
    a, _, _, _ = P.freeflyer_batch(5)
    b, _, _, _ = P.freeflyer_batch(3, first=2)
    assert np.array_equal(a[2:], b)
    gen = P.splitmix64(0x9E3779B97F4A7C15)
    v = [next(gen) for _ in range(3)]
    assert all(0.0 <= x < 1.0 for x in v) and len(set(v)) == 3
    env = P.freeflyer_env()
    assert env.shape == (14, 6)
    # the notebook builds its obstacles through Float32 (Vec3f0): min corner of the first box
    assert env[4, 0] == float(np.float32(0.460 - 0.135 - 0.05))
    boxes, sph = P.iss_corner_env(True)
    assert boxes.shape == (30, 6) and sph.shape == (2, 4)


def test_shard_bounds_cover_the_batch():
    H = g.host
    for B in (1, 7, 4096, 4097):
        for G in (1, 2, 4, 8):
            cover = []
            for r in range(G):
                lo, hi = H.shard_bounds(B, G, r)
                cover += list(range(lo, hi))
            assert cover == list(range(B))


def test_julia_wrapper_matches_the_c_header():
    """julia/GuSTOHIP*.jl cannot be executed here (no Julia toolchain): check statically what can go wrong silently at
    a ccall boundary -- every symbol it calls is exported by the library and declared in the header, and the Julia
    mirrors of the C structs have the header's fields in the header's order and types."""
    import re
    jl = "".join(open(os.path.join(ROOT, "gusto.jl_amd", "julia", f)).read() for f in ("GuSTOHIP.jl", "GuSTOHIPBatch.jl"))
    hdr = open(os.path.join(ROOT, "include", "gusto_hip.h")).read()
    syms = set(re.findall(r"ccall\(\(:(\w+), libgusto_hip\)", jl))
    assert {"gusto_create", "gusto_set_params", "gusto_set_env", "gusto_set_problems", "gusto_solve", "gusto_solve_async",
            "gusto_wait", "gusto_get_traj", "gusto_get_status", "gusto_get_history", "gusto_get_dual", "gusto_shoot",
            "gusto_create_trajopt", "gusto_set_trajopt_params", "gusto_solve_trajopt", "gusto_get_trajopt_history"} <= syms
    L = g.lib()
    for s in syms:
        assert hasattr(L, s) and re.search(r"\b%s\(" % s, hdr), s

    def c_fields(name):
        end = hdr.index("} %s;" % name)
        body = hdr[hdr.rindex("typedef struct {", 0, end) + len("typedef struct {"):end]
        body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
        out = []
        for decl in body.split(";"):
            decl = decl.strip()
            if not decl:
                continue
            ctype, rest = decl.split(None, 1)
            for v in rest.split(","):
                v = v.strip()
                dims = [int(d) if d.isdigit() else {"GUSTO_MAXN": 13, "GUSTO_MAXM": 6}[d] for d in re.findall(r"\[(\w+)\]", v)]
                out.append((re.sub(r"[\*\[].*", "", v.lstrip("*")), ctype + ("*" if v.startswith("*") else ""), int(np.prod(dims)) if dims else 0))
        return out

    def jl_fields(name):
        body = re.search(r"struct %s\b[^\n]*\n(.*?)\nend" % name, jl, re.S).group(1)
        out = []
        for f in re.split(r"[;\n]", re.sub(r"#[^\n]*", "", body)):
            f = f.strip()
            if f:
                nm, ty = f.split("::")
                out.append((nm.strip(), ty.strip()))
        return out

    def jl_type(ctype, count):
        base = {"double": "Cdouble", "int": "Cint", "double*": "Ptr{Cdouble}", "int*": "Ptr{Cint}"}[ctype]
        return f"NTuple{{{count},{base}}}" if count else base

    for cname, jname in (("gusto_scp_params", "GustoScpParams"), ("gusto_model_params", "GustoModelParams"),
                         ("gusto_history", "GustoHistory"), ("gusto_shoot_opts", "GustoShootOpts"),
                         ("gusto_trajopt_params", "GustoTrajOptParams"), ("gusto_trajopt_history", "GustoTrajOptHistory")):
        cf, jf = c_fields(cname), jl_fields(jname)
        assert [n for n, _, _ in cf] == [n for n, _ in jf], (cname, cf, jf)
        assert [jl_type(t, c) for _, t, c in cf] == [t for _, t in jf], (cname, cf, jf)
    # the robot / model scalars reach the library: no C_NULL where gusto_model_params goes
    assert "Ref{GustoModelParams}" in jl and not re.search(r"gusto_set_params.*C_NULL", jl)


def test_chain_kernel_layouts(tmp_path):
    """the wave-per-chain kernels (csrc/segw.hpp) by their compile-time layouts, without a GPU: chains partition the horizon with at
    least GUSTO_SEG_MIN_N stages each wherever launch_scp starts them, the obstacle shares partition a knot's active mask, a helper's
    LDS block holds both its [Phi Gam] double buffer and its row partials, and the workgroups fit a CU's 160 KiB -- two two-wave
    problems per CU at the BASELINE horizon (what the AUTO policy of launch.hpp assumes)."""
    import json, subprocess
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc")
    exe = str(tmp_path / "seg_layout")
    subprocess.check_call([hipcc, "--offload-arch=gfx950", "-std=c++17", "-I" + os.path.join(ROOT, "include"),
                           "-I" + os.path.join(ROOT, "gusto.jl_amd", "csrc"), os.path.join(ROOT, "tests", "c", "seg_layout.hip"), "-o", exe],
                          stderr=subprocess.DEVNULL)
    d = json.loads(subprocess.check_output([exe]).decode())
    LDS = 160 * 1024
    for name in ("astrobee_se3", "astrobee_se3_manifold"):
        m = d[name]
        for N in ("16", "33", "50", "63", "64"):
            one, two, four = m[N]
            assert two == one + m["segB2"] and four == one + m["segB4"]
            assert 8 * four <= LDS and 8 * two <= LDS, (name, N)
        assert 2 * 8 * m["50"][1] <= LDS, name                       # two two-wave problems per CU at N = 50
        assert 3 * 8 * m["50"][0] <= LDS < 4 * 8 * m["50"][0], name   # (and three one-wave problems, as DESIGN.md says)
        for hlb in (m["hlb2"], m["hlb4"]):
            assert hlb >= 2 * m["npg"] + 64 and hlb >= 64 * m["rp"]
    for key, lo in d["lo"].items():
        nch, N = map(int, key.split("_"))
        assert lo[0] == 0 and lo[-1] == N and all(b > a for a, b in zip(lo, lo[1:]))
        if N >= nch * d["min_n"]:
            assert min(b - a for a, b in zip(lo, lo[1:])) >= d["min_n"], key
        assert max(b - a for a, b in zip(lo, lo[1:])) - min(b - a for a, b in zip(lo, lo[1:])) <= 1, key
    for ns in (1, 2, 3, 4):
        masks = [int(d["share"][f"{r}_{ns}"], 16) for r in range(ns)]
        acc = 0
        for mk in masks:
            assert acc & mk == 0
            acc |= mk
        assert acc == (1 << 64) - 1
