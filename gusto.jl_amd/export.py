"""Trajectory export: the data set the reference's notebook hands to the downstream controller
(examples/freeflyerSE2.ipynb cell 6) -- group `traj` with x_traj [x_dim x N], u_traj [u_dim x N], t_traj [N]
(= collect(0:dt:Tf)) and the zero-indexed index maps ind_x / ind_u -- for one solution or a whole batch.

The notebook writes HDF5 through HDF5.jl.  A path ending in .h5 gets a real HDF5 file (h5lite.py writes the container
directly to the file format specification -- the Python of this image has no HDF5 library; libhdf5's h5dump / h5ls
read the result) with the data set layout HDF5.jl produces for the notebook's arrays: x_traj is stored with HDF5
dimensions (N, x_dim), which a column-major reader sees as the notebook's x_dim x N matrix.  The same tree can also be
written as a MATLAB v5 file (.mat: scipy.io.savemat; MAT.jl, which the reference already uses for its environments --
iss_corner.jl:11 -- reads it) or as .npz with '/'-joined keys; there the arrays are stored state-index first.
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


def _tree(model_id, X, U, tf, extra=None, column_major_reader=False):
    """X [.., N, n], U [.., N, m] in the C ABI layout -> the notebook's layout (state index first, knot second).
    column_major_reader: keep the C layout [.., N, n] -- HDF5 dimensions are row-major, so HDF5.jl / MATLAB present such
    a data set as n x N (x B), exactly the matrix the notebook wrote."""
    X, U = np.asarray(X, float), np.asarray(U, float)
    N = X.shape[-2]
    tf = np.asarray(tf, float)
    t = np.linspace(0.0, 1.0, N) * tf[..., None] if tf.ndim else np.linspace(0.0, float(tf), N)
    names_x, names_u = INDEX_MAPS[model_id]
    sw = (lambda a: a) if column_major_reader else (lambda a: np.swapaxes(a, -1, -2))
    out = {"traj": {"x_traj": sw(X), "u_traj": sw(U), "t_traj": t},
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
    """path ending in .h5 -> HDF5 (the notebook's container); .mat -> MATLAB v5 with nested structs; .npz -> flat keys
    'traj/x_traj', ...  Returns the tree that was written."""
    tree = _tree(model_id, X, U, tf, status, column_major_reader=path.endswith((".h5", ".hdf5")))
    if path.endswith((".h5", ".hdf5")):
        from . import h5lite
        h5lite.write_h5(path, tree)
    elif path.endswith(".mat"):
        import scipy.io
        scipy.io.savemat(path, tree, do_compression=True)
    elif path.endswith(".npz"):
        np.savez_compressed(path, **_flatten(tree))
    else:
        raise ValueError("export.write: path must end in .h5, .mat or .npz")
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
    """Round trip of write() for .mat / .npz (tests, and the downstream consumer's view of the file).  An .h5 file is read
    with an HDF5 library (HDF5.jl: h5read(path, "traj/x_traj"); tests/h5read.py is the spec-following reader of the suite)."""
    if path.endswith((".h5", ".hdf5")):
        raise ValueError("export.read: open .h5 files with an HDF5 library")
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
