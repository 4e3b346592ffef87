"""ctypes binding of libgusto_hip.so (include/gusto_hip.h).  No fallback: if the HIP library is missing or no
GPU is present the functions raise -- the product never routes through a CPU path."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(_HERE)
LIB_PATH = os.path.join(_HERE, "libgusto_hip.so")
MAXN, MAXM = 13, 6

FREEFLYER_SE2, DUBINS_CAR, ASTROBEE_SE3, ASTROBEE_SE3_MANIFOLD = 0, 1, 2, 3
MODEL_DIMS = {0: (6, 3), 1: (3, 1), 2: (12, 6), 3: (13, 6)}
SCP_STATUS = {0: "NA", 1: "OK", 2: "InaccurateModel", 3: "ViolatesConstraints", 4: "TrustRegionViolated"}
SOLVER_STATUS = {0: "NA", 1: "OPTIMAL", 2: "ALMOST_LOCALLY_SOLVED", 3: "FAILED"}
STOP_REASON = {0: "MaxIter", 1: "Converged", 2: "SubproblemFailed", 3: "OmegaMaxExceeded", 4: "HistoryFull"}

# every symbol include/gusto_hip.h declares
SYMBOLS = ["gusto_default_params", "gusto_default_ipm_opts", "gusto_model_dims", "gusto_create", "gusto_destroy",
           "gusto_last_error", "gusto_set_params", "gusto_set_ipm_opts", "gusto_set_env", "gusto_set_env_batch", "gusto_set_schedule", "gusto_set_decomposition", "gusto_dev_workspace_bytes",
           "gusto_set_stream",
           "gusto_set_problems", "gusto_set_problems_dev", "gusto_solve", "gusto_solve_async", "gusto_set_active", "gusto_wait",
           "gusto_last_solve_ms", "gusto_get_traj",
           "gusto_get_traj_dev", "gusto_gather_peer", "gusto_get_status", "gusto_get_dual", "gusto_get_history", "gusto_get_hist_cap",
           "gusto_set_trust_state", "gusto_subproblem", "gusto_default_shoot_opts", "gusto_shoot", "gusto_get_shoot",
           "gusto_default_trajopt_params", "gusto_create_trajopt", "gusto_set_trajopt_params", "gusto_solve_trajopt", "gusto_solve_trajopt_async",
           "gusto_get_trajopt_history", "gusto_subproblem_trajopt",
           "gusto_dev_get_prof", "gusto_dev_launch_info"]


class ScpParams(C.Structure):
    _fields_ = [(k, C.c_double) for k in
                ("Delta0", "omega0", "omega_max", "eps", "rho0", "rho1", "beta_succ", "beta_fail", "gamma_fail",
                 "convergence_threshold")]


class ModelParams(C.Structure):
    _fields_ = [("mass", C.c_double), ("Jdiag", C.c_double * 3), ("radius", C.c_double), ("clearance", C.c_double),
                ("hard_limit_vel", C.c_double), ("hard_limit_accel", C.c_double), ("hard_limit_omega", C.c_double),
                ("hard_limit_alpha", C.c_double), ("dubins_v", C.c_double), ("dubins_k", C.c_double),
                ("u_max", C.c_double), ("u_min", C.c_double), ("x_max", C.c_double * MAXN),
                ("x_min", C.c_double * MAXN), ("n_robot_comp", C.c_int), ("comp_off", (C.c_double * 3) * 2)]


class IpmOpts(C.Structure):
    _fields_ = [("tol", C.c_double), ("tol_acc", C.c_double), ("mu_floor", C.c_double), ("tr_tol", C.c_double),
                ("mu_warm", C.c_double), ("max_iter", C.c_int), ("acc_iter", C.c_int), ("mu_warm_gain", C.c_double), ("mu_warm_max", C.c_double), ("sigma_max", C.c_double)]


class ShootOpts(C.Structure):
    _fields_ = [("substeps", C.c_int), ("max_newton", C.c_int), ("ftol", C.c_double), ("no_group_pass", C.c_int)]


class TrajOptParams(C.Structure):
    """gusto_trajopt_params = SCPParam_TrajOpt (scp_trajopt.jl:3-30)"""
    _fields_ = [(k, C.c_double) for k in ("mu0", "s0", "c", "tau_plus", "tau_minus", "k", "ftol", "xtol", "ctol")] + \
               [(k, C.c_int) for k in ("max_penalty_iteration", "max_convex_iteration", "max_trust_iteration")]


class TrajOptHistory(C.Structure):
    _fields_ = [("hist_cap", C.c_int)] + [(k, C.c_void_p) for k in
                ("n_solves", "n_mu", "n_xtol", "n_ftol", "n_ctol", "rho_vec", "s_vec", "mu_vec", "xtol_vec", "ftol_vec",
                 "ctol_vec", "J_true", "J_full", "convergence_measure", "solver_status", "ipm_iters")]


class History(C.Structure):
    _fields_ = [("hist_cap", C.c_int), ("n_hist", C.c_void_p), ("nJ", C.c_void_p), ("n_rho", C.c_void_p),
                ("J_true", C.c_void_p), ("J_full", C.c_void_p), ("convergence_measure", C.c_void_p),
                ("Delta", C.c_void_p), ("omega", C.c_void_p), ("rho", C.c_void_p), ("accept_solution", C.c_void_p),
                ("scp_status", C.c_void_p), ("solver_status", C.c_void_p), ("trust_region_satisfied", C.c_void_p),
                ("convex_ineq_satisfied", C.c_void_p), ("ipm_iters", C.c_void_p)]


def build(force=False, verbose=False):
    """Compile libgusto_hip.so for gfx950 with hipcc (cross-compiles without a GPU): one translation unit per
    model plus the C ABI, compiled in parallel, then linked."""
    from concurrent.futures import ThreadPoolExecutor
    csrc = os.path.join(_HERE, "csrc")
    deps = [os.path.join(csrc, f) for f in os.listdir(csrc)] + [os.path.join(_ROOT, "include", "gusto_hip.h")]
    if not force and os.path.exists(LIB_PATH) and all(os.path.getmtime(s) <= os.path.getmtime(LIB_PATH) for s in deps):
        return LIB_PATH
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    flags = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-I" + os.path.join(_ROOT, "include"), "-fPIC",
             "-Wno-unused-value", "-Wno-pass-failed"]
    units = ["gusto_hip", "shoot", "model_0", "model_1", "model_2", "model_3", "model_4", "model_5", "model_6"]
    bdir = os.path.join(_HERE, "build")
    os.makedirs(bdir, exist_ok=True)

    def cc(u):
        cmd = [hipcc] + flags + ["-c", os.path.join(csrc, u + ".hip"), "-o", os.path.join(bdir, u + ".o")]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)

    with ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 1)) as ex:
        list(ex.map(cc, units))
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC"] + [os.path.join(bdir, u + ".o") for u in units] + \
          ["-o", LIB_PATH]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return LIB_PATH


_lib = None
_dp = np.ctypeslib.ndpointer(dtype=np.float64, flags="C_CONTIGUOUS")


def lib():
    """Load the HIP library; raises if it has not been built (no CPU fallback exists)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(f"{LIB_PATH} is missing: build it with __graft_entry__.build() "
                               "(hipcc --offload-arch=gfx950); gusto.jl_amd has no CPU fallback")
        L = C.CDLL(LIB_PATH)
        vp, ci = C.c_void_p, C.c_int
        L.gusto_last_error.restype = C.c_char_p
        L.gusto_last_error.argtypes = [vp]
        L.gusto_default_params.argtypes = [ci, C.POINTER(ScpParams), C.POINTER(ModelParams)]
        L.gusto_default_ipm_opts.argtypes = [C.POINTER(IpmOpts)]
        L.gusto_model_dims.argtypes = [ci, C.POINTER(ci), C.POINTER(ci)]
        L.gusto_create.argtypes = [C.POINTER(vp), ci, ci, ci, ci, ci]
        L.gusto_destroy.argtypes = [vp]
        L.gusto_dev_launch_info.argtypes = [vp, C.POINTER(ci), C.POINTER(ci), C.POINTER(ci)]
        L.gusto_dev_workspace_bytes.argtypes = [vp, C.POINTER(C.c_longlong)]
        L.gusto_set_params.argtypes = [vp, C.POINTER(ScpParams), C.POINTER(ModelParams)]
        L.gusto_set_ipm_opts.argtypes = [vp, C.POINTER(IpmOpts)]
        L.gusto_set_env.argtypes = [vp, ci, vp, ci, vp]
        L.gusto_set_env_batch.argtypes = [vp, ci, vp, vp, vp, vp]
        L.gusto_set_stream.argtypes = [vp, vp]
        L.gusto_set_problems.argtypes = [vp, ci, vp, vp, vp, vp, vp, vp]
        L.gusto_set_problems_dev.argtypes = [vp, ci, vp, vp, vp, vp, vp, vp]
        L.gusto_set_schedule.argtypes = [vp, ci, ci]
        L.gusto_set_decomposition.argtypes = [vp, ci]
        L.gusto_solve.argtypes = [vp, ci, ci]
        L.gusto_solve_async.argtypes = [vp, ci, ci]
        L.gusto_wait.argtypes = [vp]
        L.gusto_set_active.argtypes = [vp, vp]
        L.gusto_last_solve_ms.argtypes = [vp, C.POINTER(C.c_double)]
        L.gusto_get_traj.argtypes = [vp, vp, vp]
        L.gusto_get_traj_dev.argtypes = [vp, C.POINTER(vp), C.POINTER(vp)]
        L.gusto_get_status.argtypes = [vp, vp, vp, vp, vp, vp]
        L.gusto_gather_peer.argtypes = [vp, ci, C.POINTER(vp), C.POINTER(vp), C.POINTER(vp), vp, vp, C.POINTER(ci)]
        L.gusto_get_dual.argtypes = [vp, vp]
        L.gusto_get_history.argtypes = [vp, C.POINTER(History)]
        L.gusto_get_hist_cap.argtypes = [vp, C.POINTER(ci)]
        L.gusto_set_trust_state.argtypes = [vp, vp, vp]
        L.gusto_default_shoot_opts.argtypes = [C.POINTER(ShootOpts)]
        L.gusto_shoot.argtypes = [vp, vp, C.POINTER(ShootOpts)]
        L.gusto_get_shoot.argtypes = [vp, vp, vp, vp, vp, vp, vp]
        L.gusto_subproblem.argtypes = [vp, ci, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp]
        L.gusto_default_trajopt_params.argtypes = [ci, C.POINTER(TrajOptParams)]
        L.gusto_create_trajopt.argtypes = [C.POINTER(vp), ci, ci, ci, ci, ci]
        L.gusto_set_trajopt_params.argtypes = [vp, C.POINTER(TrajOptParams)]
        L.gusto_solve_trajopt.argtypes = [vp, ci]
        L.gusto_solve_trajopt_async.argtypes = [vp, ci]
        L.gusto_get_trajopt_history.argtypes = [vp, C.POINTER(TrajOptHistory)]
        L.gusto_subproblem_trajopt.argtypes = [vp, ci, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp]
        _lib = L
    return _lib


def default_params(model):
    sp, mp = ScpParams(), ModelParams()
    rc = lib().gusto_default_params(model, C.byref(sp), C.byref(mp))
    if rc:
        raise ValueError(f"gusto_default_params({model}) -> {rc}")
    return sp, mp


def default_trajopt_params(model):
    tp = TrajOptParams()
    rc = lib().gusto_default_trajopt_params(model, C.byref(tp))
    if rc:
        raise ValueError(f"gusto_default_trajopt_params({model}) -> {rc}: the model has no TrajOpt variant")
    return tp


def default_ipm_opts():
    o = IpmOpts()
    lib().gusto_default_ipm_opts(C.byref(o))
    return o


def _arr(a, dtype=np.float64):
    return np.ascontiguousarray(np.asarray(a, dtype=dtype))


class GustoError(RuntimeError):
    pass


class BatchSolver:
    """Thin owner of one gusto_handle: a batch of SCP problems of one model on one GPU."""
    # gusto_set_decomposition applied to every new handle of the models it means something for (0 = the library's choice); tests set
    # it: 2 (a lane per problem) reaches dubins_car handles, 1 / 3 / 4 (one / two / four waves per problem) the 12/13-state models
    default_decomposition = 0

    def __init__(self, model, N, batch_cap, hist_cap=64, device=0, boxes=None, spheres=None, scp_params=None,
                 model_params=None, ipm_opts=None):
        self.L = lib()
        self.model, self.N, self.batch_cap, self.hist_cap, self.device = model, N, batch_cap, hist_cap, device
        self.n, self.m = MODEL_DIMS[model]
        self.h = C.c_void_p()
        rc = self._create(model, N, batch_cap, hist_cap, device)
        if rc:
            msg = self.L.gusto_last_error(self.h if self.h else None)
            self.h = C.c_void_p()
            raise GustoError(f"gusto_create -> {rc}: {msg.decode() if msg else ''}")
        self.B = 0
        if scp_params is not None or model_params is not None:
            self._chk(self.L.gusto_set_params(self.h, C.byref(scp_params) if scp_params is not None else None,
                                              C.byref(model_params) if model_params is not None else None), "set_params")
        if ipm_opts is not None:
            self._chk(self.L.gusto_set_ipm_opts(self.h, C.byref(ipm_opts)), "set_ipm_opts")
        if BatchSolver.default_decomposition and type(self) is BatchSolver:
            if (model == DUBINS_CAR) if BatchSolver.default_decomposition == 2 else (model in (ASTROBEE_SE3, ASTROBEE_SE3_MANIFOLD)):
                self.set_decomposition(BatchSolver.default_decomposition)
        self.set_env(boxes, spheres)

    def _create(self, model, N, batch_cap, hist_cap, device):
        return self.L.gusto_create(C.byref(self.h), model, N, batch_cap, hist_cap, device)

    def _chk(self, rc, what):
        if rc:
            msg = self.L.gusto_last_error(self.h)
            raise GustoError(f"gusto_{what} -> {rc}: {msg.decode() if msg else ''}")

    def close(self):
        if getattr(self, "h", None) and self.h:
            self.L.gusto_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_env(self, boxes=None, spheres=None):
        self.boxes = _arr(boxes if boxes is not None else np.zeros((0, 6))).reshape(-1, 6)
        self.spheres = _arr(spheres if spheres is not None else np.zeros((0, 4))).reshape(-1, 4)
        self._chk(self.L.gusto_set_env(self.h, len(self.boxes), self.boxes.ctypes.data, len(self.spheres),
                                       self.spheres.ctypes.data), "set_env")

    def set_env_batch(self, boxes_list, spheres_list=None):
        """gusto_set_env_batch: one keep-out set per problem -- boxes_list[b] is [n_box_b, 6], spheres_list[b] [n_sph_b, 4]
        (None = no spheres anywhere).  In the reference every ProblemDefinition owns its env (types.jl:32-39)."""
        B = len(boxes_list)
        bl = [_arr(b if b is not None else np.zeros((0, 6))).reshape(-1, 6) for b in boxes_list]
        sl = [_arr(s if s is not None else np.zeros((0, 4))).reshape(-1, 4) for s in (spheres_list or [None] * B)]
        if len(sl) != B:
            raise ValueError("set_env_batch: as many sphere tables as box tables")
        nb, ns = _arr([len(b) for b in bl], np.int32), _arr([len(s) for s in sl], np.int32)
        box = _arr(np.concatenate(bl, axis=0)) if bl else np.zeros((0, 6))
        sph = _arr(np.concatenate(sl, axis=0)) if sl else np.zeros((0, 4))
        self.boxes, self.spheres = bl, sl
        self._chk(self.L.gusto_set_env_batch(self.h, B, nb.ctypes.data, box.ctypes.data, ns.ctypes.data, sph.ctypes.data),
                  "set_env_batch")

    def set_problems(self, x_init, goal_lo, goal_hi, tf, X0=None, U0=None):
        x_init, goal_lo, goal_hi = _arr(x_init).reshape(-1, self.n), _arr(goal_lo).reshape(-1, self.n), \
            _arr(goal_hi).reshape(-1, self.n)
        B = x_init.shape[0]
        tf = _arr(np.broadcast_to(np.asarray(tf, dtype=np.float64), (B,)))
        self._keep = (x_init, goal_lo, goal_hi, tf, None if X0 is None else _arr(X0), None if U0 is None else _arr(U0))
        self._chk(self.L.gusto_set_problems(self.h, B, x_init.ctypes.data, goal_lo.ctypes.data, goal_hi.ctypes.data,
                                            tf.ctypes.data, None if X0 is None else self._keep[4].ctypes.data,
                                            None if U0 is None else self._keep[5].ctypes.data), "set_problems")
        self.B = B

    def set_trust_state(self, Delta=None, omega=None):
        """gusto_set_trust_state: the caller's own Delta_vec[end] / omega_vec[end] for the next trip of every problem."""
        D = None if Delta is None else _arr(np.broadcast_to(np.asarray(Delta, dtype=np.float64), (self.B,)))
        W = None if omega is None else _arr(np.broadcast_to(np.asarray(omega, dtype=np.float64), (self.B,)))
        self._chk(self.L.gusto_set_trust_state(self.h, None if D is None else D.ctypes.data,
                                               None if W is None else W.ctypes.data), "set_trust_state")

    def set_decomposition(self, decomposition):
        """0 auto, 1 one wave per problem, 2 a lane per problem (dubins_car, -DGUSTO_WITH_LANE builds), 3 / 4 two / four waves per
        problem (astrobeeSE3, astrobeeSE3manifold: csrc/segw.hpp); gusto_hip.h: gusto_set_decomposition."""
        self._chk(self.L.gusto_set_decomposition(self.h, int(decomposition)), "set_decomposition")

    def set_schedule(self, probe_iters=2, min_batch=2048):
        self._chk(self.L.gusto_set_schedule(self.h, int(probe_iters), int(min_batch)), "set_schedule")

    def solve(self, max_iter=30, force=False):
        self._chk(self.L.gusto_solve(self.h, int(max_iter), int(bool(force))), "solve")

    def set_active(self, active=None):
        """gusto_set_active: the problems the following solve / shoot calls work on (boolean mask [B]; None = all)."""
        if active is None:
            self._chk(self.L.gusto_set_active(self.h, None), "set_active")
            return
        a = _arr(np.asarray(active).astype(bool), np.int32).reshape(self.B)
        self._chk(self.L.gusto_set_active(self.h, a.ctypes.data), "set_active")

    def solve_async(self, max_iter=30, force=False):
        """Enqueue the solve on the handle's stream and return; wait() (or any getter) completes it."""
        self._chk(self.L.gusto_solve_async(self.h, int(max_iter), int(bool(force))), "solve_async")

    def wait(self):
        self._chk(self.L.gusto_wait(self.h), "wait")

    def set_problems_dev(self, B, x_init_ptr, goal_lo_ptr, goal_hi_ptr, tf_ptr, X0_ptr=None, U0_ptr=None):
        """gusto_set_problems_dev: inputs already resident in HBM (raw device pointers)."""
        self._chk(self.L.gusto_set_problems_dev(self.h, int(B), x_init_ptr, goal_lo_ptr, goal_hi_ptr, tf_ptr, X0_ptr,
                                                U0_ptr), "set_problems_dev")
        self.B = int(B)

    def last_solve_ms(self):
        ms = C.c_double()
        self._chk(self.L.gusto_last_solve_ms(self.h, C.byref(ms)), "last_solve_ms")
        return ms.value

    def traj(self):
        X, U = np.zeros((self.B, self.N, self.n)), np.zeros((self.B, self.N, self.m))
        self._chk(self.L.gusto_get_traj(self.h, X.ctypes.data, U.ctypes.data), "get_traj")
        return X, U

    def traj_dev(self):
        """gusto_get_traj_dev as zero-copy torch views of the handle's HBM buffers: X [B,N,n], U [B,N,m] (valid until
        the next set_problems / solve on this handle).  For device-side consumers, e.g. the RCCL gather."""
        import torch
        px, pu = C.c_void_p(), C.c_void_p()
        self._chk(self.L.gusto_get_traj_dev(self.h, C.byref(px), C.byref(pu)), "get_traj_dev")

        class _View:       # __cuda_array_interface__ v2: lets torch wrap a raw device pointer without a copy
            def __init__(self, ptr, shape):
                self.__cuda_array_interface__ = dict(shape=shape, typestr="<f8", data=(int(ptr), False), version=2)

        dev = torch.device("cuda", self.device)
        return (torch.as_tensor(_View(px.value, (self.B, self.N, self.n)), device=dev),
                torch.as_tensor(_View(pu.value, (self.B, self.N, self.m)), device=dev))

    def gather_peer(self, sources, host=True):
        """gusto_gather_peer: the shards of `sources` (BatchSolvers, one per GPU, solves possibly still in flight) onto this
        handle's GPU by direct peer copies; returns (X, U) of all problems in the order of `sources` -- numpy arrays, or
        with host=False zero-copy torch views of the gathered device buffers."""
        hs = (C.c_void_p * len(sources))(*[q.h for q in sources])
        px, pu, bt = C.c_void_p(), C.c_void_p(), C.c_int()
        Bt = sum(q.B for q in sources)
        X = np.zeros((Bt, self.N, self.n)) if host else None
        U = np.zeros((Bt, self.N, self.m)) if host else None
        self._chk(self.L.gusto_gather_peer(self.h, len(sources), hs, C.byref(px), C.byref(pu),
                                           X.ctypes.data if host else None, U.ctypes.data if host else None, C.byref(bt)),
                  "gather_peer")
        assert bt.value == Bt
        if host:
            return X, U
        import torch

        class _View:
            def __init__(self, ptr, shape):
                self.__cuda_array_interface__ = dict(shape=shape, typestr="<f8", data=(int(ptr), False), version=2)

        dev = torch.device("cuda", self.device)
        return (torch.as_tensor(_View(px.value, (Bt, self.N, self.n)), device=dev),
                torch.as_tensor(_View(pu.value, (Bt, self.N, self.m)), device=dev))

    def status(self):
        a = [np.zeros(self.B, dtype=np.int32) for _ in range(5)]
        self._chk(self.L.gusto_get_status(self.h, *[x.ctypes.data for x in a]), "get_status")
        return dict(iterations=a[0], converged=a[1].astype(bool), successful=a[2].astype(bool), stop_reason=a[3],
                    ipm_iters=a[4])

    def dual(self):
        d = np.zeros((self.B, self.n))
        self._chk(self.L.gusto_get_dual(self.h, d.ctypes.data), "get_dual")
        return d

    def history(self):
        B, H = self.B, self.hist_cap
        dk = ("J_true", "J_full", "convergence_measure", "Delta", "omega", "rho")
        ik = ("accept_solution", "scp_status", "solver_status", "trust_region_satisfied", "convex_ineq_satisfied",
              "ipm_iters")
        out = {k: np.zeros((B, H)) for k in dk}
        out.update({k: np.zeros((B, H), dtype=np.int32) for k in ik})
        cnt = {k: np.zeros(B, dtype=np.int32) for k in ("n_hist", "nJ", "n_rho")}
        hs = History()
        hs.hist_cap = H
        for k, v in list(out.items()) + list(cnt.items()):
            setattr(hs, k, v.ctypes.data)
        self._chk(self.L.gusto_get_history(self.h, C.byref(hs)), "get_history")
        out.update(cnt)
        return out

    def launch_info(self):
        """gusto_dev_launch_info: (persistent workgroups, LDS bytes per workgroup, workgroups per CU) of the last launch."""
        a, b, c = C.c_int(), C.c_int(), C.c_int()
        self._chk(self.L.gusto_dev_launch_info(self.h, C.byref(a), C.byref(b), C.byref(c)), "dev_launch_info")
        return a.value, b.value, c.value

    def workspace_bytes(self):
        """gusto_dev_workspace_bytes: bytes of the handle's interior point workspace in HBM."""
        v = C.c_longlong()
        self._chk(self.L.gusto_dev_workspace_bytes(self.h, C.byref(v)), "dev_workspace_bytes")
        return int(v.value)

    def shoot(self, p0=None, substeps=4, max_newton=100, ftol=1e-3, group_pass=True):
        """gusto_shoot + gusto_get_shoot: indirect shooting of every problem from p0 (default: the SCP duals)."""
        o = ShootOpts(substeps=substeps, max_newton=max_newton, ftol=ftol, no_group_pass=int(not group_pass))
        pv = None if p0 is None else _arr(p0).reshape(self.B, self.n)
        self._chk(self.L.gusto_shoot(self.h, None if pv is None else pv.ctypes.data, C.byref(o)), "shoot")
        B = self.B
        st, it = np.zeros(B, dtype=np.int32), np.zeros(B, dtype=np.int32)
        res, pp = np.zeros(B), np.zeros((B, self.n))
        X, U = np.zeros((B, self.N, self.n)), np.zeros((B, self.N, self.m))
        self._chk(self.L.gusto_get_shoot(self.h, st.ctypes.data, it.ctypes.data, res.ctypes.data, pp.ctypes.data,
                                         X.ctypes.data, U.ctypes.data), "get_shoot")
        return dict(status=st, newton_iters=it, resid=res, p0=pp, X=X, U=U)

    def subproblem(self, Xp, Up, Delta, omega, toggle):
        B = self.B
        Xp, Up = _arr(Xp).reshape(B, self.N, self.n), _arr(Up).reshape(B, self.N, self.m)
        Delta, omega, toggle = (_arr(np.broadcast_to(np.asarray(v, dtype=np.float64), (B,))) for v in
                                (Delta, omega, toggle))
        Xn, Un, obj = np.zeros_like(Xp), np.zeros_like(Up), np.zeros(B)
        st, it = np.zeros(B, dtype=np.int32), np.zeros(B, dtype=np.int32)
        self._chk(self.L.gusto_subproblem(self.h, B, Xp.ctypes.data, Up.ctypes.data, Delta.ctypes.data,
                                          omega.ctypes.data, toggle.ctypes.data, Xn.ctypes.data, Un.ctypes.data,
                                          obj.ctypes.data, st.ctypes.data, it.ctypes.data), "subproblem")
        return dict(X=Xn, U=Un, obj=obj, status=st, iters=it, dual=self.dual())


class TrajOptSolver(BatchSolver):
    """A gusto_handle created by gusto_create_trajopt: the TrajOpt algorithm (src/scp/scp_trajopt.jl) for a batch of problems
    of FreeflyerSE2, AstrobeeSE3 or AstrobeeSE3Manifold.  set_env / set_problems / traj / status / dual / last_solve_ms as for BatchSolver."""

    def __init__(self, model, N, batch_cap, hist_cap=272, device=0, boxes=None, spheres=None, model_params=None,
                 trajopt_params=None, ipm_opts=None):
        super().__init__(model, N, batch_cap, hist_cap, device, boxes, spheres, None, model_params, ipm_opts)
        if trajopt_params is not None:
            self._chk(self.L.gusto_set_trajopt_params(self.h, C.byref(trajopt_params)), "set_trajopt_params")

    def _create(self, model, N, batch_cap, hist_cap, device):
        return self.L.gusto_create_trajopt(C.byref(self.h), model, N, batch_cap, hist_cap, device)

    def solve(self, max_iter=125, force=False):
        self._chk(self.L.gusto_solve_trajopt(self.h, int(max_iter)), "solve_trajopt")

    def solve_async(self, max_iter=125, force=False):
        """gusto_solve_trajopt_async: enqueue the launch of the batch and return; wait() or any getter completes it."""
        self._chk(self.L.gusto_solve_trajopt_async(self.h, int(max_iter)), "solve_trajopt_async")

    def history(self):
        B, H = self.B, self.hist_cap
        dk = ("rho_vec", "s_vec", "mu_vec", "xtol_vec", "ftol_vec", "ctol_vec", "J_true", "J_full", "convergence_measure")
        out = {k: np.zeros((B, H)) for k in dk}
        out.update({k: np.zeros((B, H), dtype=np.int32) for k in ("solver_status", "ipm_iters")})
        cnt = {k: np.zeros(B, dtype=np.int32) for k in ("n_solves", "n_mu", "n_xtol", "n_ftol", "n_ctol")}
        hs = TrajOptHistory()
        hs.hist_cap = H
        for k, v in list(out.items()) + list(cnt.items()):
            setattr(hs, k, v.ctypes.data)
        self._chk(self.L.gusto_get_trajopt_history(self.h, C.byref(hs)), "get_trajopt_history")
        out.update(cnt)
        return out

    def subproblem(self, Xp, Up, mu, s):
        """gusto_subproblem_trajopt: one convex subproblem per problem around (Xp, Up) with penalty mu and trust region s."""
        B = self.B
        Xp, Up = _arr(Xp).reshape(B, self.N, self.n), _arr(Up).reshape(B, self.N, self.m)
        mu, s = (_arr(np.broadcast_to(np.asarray(v, dtype=np.float64), (B,))) for v in (mu, s))
        Xn, Un, Dn, obj = np.zeros_like(Xp), np.zeros_like(Up), np.zeros_like(Xp), np.zeros(B)
        st, it = np.zeros(B, dtype=np.int32), np.zeros(B, dtype=np.int32)
        self._chk(self.L.gusto_subproblem_trajopt(self.h, B, Xp.ctypes.data, Up.ctypes.data, mu.ctypes.data, s.ctypes.data,
                                                  Xn.ctypes.data, Un.ctypes.data, Dn.ctypes.data, obj.ctypes.data,
                                                  st.ctypes.data, it.ctypes.data), "subproblem_trajopt")
        return dict(X=Xn, U=Un, D=Dn, obj=obj, status=st, iters=it, dual=self.dual())

    def shoot(self, *a, **k):
        raise GustoError("TrajOptSolver: shooting belongs to the GuSTO path")
