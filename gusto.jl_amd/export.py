"""Trajectory export: the data set the reference's notebook hands to the downstream controller
(examples/freeflyerSE2.ipynb cell 6) -- group `traj` with x_traj [x_dim x N], u_traj [u_dim x N], t_traj [N]
(= collect(0:dt:Tf)) and the zero-indexed index maps ind_x / ind_u -- for one solution or a whole batch.

The notebook writes HDF5 through HDF5.jl; this image has no HDF5 library (no h5py, no libhdf5), so the same groups
and data set names are written as a MATLAB v5 file (scipy.io.savemat; MAT.jl, which the reference already uses for
its environments -- iss_corner.jl:11 -- reads it: matread(path)["traj"]["x_traj"]) and as .npz with '/'-joined keys.
A batch adds the leading problem axis and the per-problem status vectors."""
import numpy as np

from . import _capi

# notebook cell 6: ind_x = {x, y, theta, vx, vy, omega}, ind_u = {Fx, Fy, M}; the other models by their state layout
INDEX_MAPS = {
    _capi.FREEFLYER_SE2: (("x", "y", "theta", "vx", "vy", "omega"), ("Fx", "Fy", "M")),
    _capi.DUBINS_CAR: (("x", "y", "theta"), ("u",)),
    _capi.ASTROBEE_SE3: (("rx", "ry", "rz", "vx", "vy", "vz", "px", "py", "pz", "wx", "wy", "wz"),
                         ("Fx", "Fy", "Fz", "Mx", "My", "Mz")),
    _capi.ASTROBEE_SE3_MANIFOLD: (("rx", "ry", "rz", "vx", "vy", "vz", "qw", "qx", "qy", "qz", "wx", "wy", "wz"),
                                  ("Fx", "Fy", "Fz", "Mx", "My", "Mz")),
}


def _tree(model_id, X, U, tf, extra=None):
    """X [.., N, n], U [.., N, m] in the C ABI layout -> the notebook's layout (state index first, knot second)."""
    X, U = np.asarray(X, float), np.asarray(U, float)
    N = X.shape[-2]
    tf = np.asarray(tf, float)
    t = np.linspace(0.0, 1.0, N) * tf[..., None] if tf.ndim else np.linspace(0.0, float(tf), N)
    names_x, names_u = INDEX_MAPS[model_id]
    out = {"traj": {"x_traj": np.swapaxes(X, -1, -2), "u_traj": np.swapaxes(U, -1, -2), "t_traj": t},
           "ind_x": {k: np.int64(i) for i, k in enumerate(names_x)},
           "ind_u": {k: np.int64(i) for i, k in enumerate(names_u)}}
    if extra:
        out["status"] = {k: np.asarray(v) for k, v in extra.items()}
    return out


def _flatten(tree, prefix=""):
    flat = {}
    for k, v in tree.items():
        if isinstance(v, dict):
            flat.update(_flatten(v, prefix + k + "/"))
        else:
            flat[prefix + k] = v
    return flat


def write(path, model_id, X, U, tf, status=None):
    """path ending in .mat -> MATLAB v5 with nested structs; .npz -> flat keys 'traj/x_traj', ...  Returns the tree."""
    tree = _tree(model_id, X, U, tf, status)
    if path.endswith(".mat"):
        import scipy.io
        scipy.io.savemat(path, tree, do_compression=True)
    elif path.endswith(".npz"):
        np.savez_compressed(path, **_flatten(tree))
    else:
        raise ValueError("export.write: path must end in .mat or .npz (no HDF5 library in this environment)")
    return tree


def write_solution(path, TOS):
    """One TrajectoryOptimizationSolution (host mirror): TOS.traj.X is already x_dim x N as in the reference."""
    model_id = TOS.SCPS.SCPP.PD.model.model_id
    return write(path, model_id, TOS.traj.X.T, TOS.traj.U.T, TOS.traj.Tf,
                 dict(converged=TOS.SCPS.converged, successful=TOS.SCPS.successful, iterations=TOS.SCPS.iterations))


def write_batch(path, solver, tf):
    """Everything a BatchSolver holds after a solve: X [B][n][N], U [B][m][N], t [B][N] + per-problem status."""
    X, U = solver.traj()
    st = solver.status()
    return write(path, solver.model, X, U, np.asarray(tf, float),
                 dict(converged=st["converged"].astype(np.int8), successful=st["successful"].astype(np.int8),
                      iterations=st["iterations"], stop_reason=st["stop_reason"]))


def read(path):
    """Round trip of write() (tests, and the downstream consumer's view of the file)."""
    if path.endswith(".mat"):
        import scipy.io
        m = scipy.io.loadmat(path, simplify_cells=True)
        return {k: v for k, v in m.items() if not k.startswith("__")}
    flat = dict(np.load(path))
    tree = {}
    for k, v in flat.items():
        d = tree
        parts = k.split("/")
        for p in parts[:-1]:
            d = d.setdefault(p, {})
        d[parts[-1]] = v
    return tree
