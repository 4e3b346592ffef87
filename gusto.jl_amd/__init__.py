"""gusto.jl_amd -- MI355X-native batched GuSTO sequential convex programming.

The package holds the HIP kernels + C ABI (csrc/, libgusto_hip.so) and the host-side mirror of the reference's
driver interface (host.py: solve_SCP!, solve_gusto_hip!).  Import it through the `gusto_jl_amd` shim at the
repository root (the directory name contains a dot)."""
from . import _capi, problems  # noqa: F401
from ._capi import (ASTROBEE_SE3, ASTROBEE_SE3_MANIFOLD, DUBINS_CAR, FREEFLYER_SE2, BatchSolver, GustoError,  # noqa: F401
                    IpmOpts, MODEL_DIMS, ModelParams, ScpParams, TrajOptParams, TrajOptSolver, build, default_ipm_opts,
                    default_params, default_trajopt_params, lib)
from . import host  # noqa: F401,E402
from . import export  # noqa: F401,E402
