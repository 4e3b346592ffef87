"""Problem definitions of the benchmark configs: environments as static obstacle tables
and reproducible random initial states.  Mirrors what the reference's notebooks build with Table(:stanford),
HyperRectangle obstacles and ProblemDefinition (examples/freeflyerSE2.ipynb cell 2, src/environment/table.jl)."""
import numpy as np

FT2M = 0.3048


def table_stanford_boxes():
    """Table(:stanford) keep-out slabs (src/environment/table.jl:11-55) as AABBs [min xyz | max xyz]."""
    lo = np.array([0.0, 0.0, 0.0])
    hi = np.array([12.0, 9.0, 0.001]) * FT2M
    a = 10.0
    b = [([hi[0], -a, -a], [hi[0] + a, a, a]), ([lo[0] - a, -a, -a], [lo[0], a, a]),
         ([-a, hi[1], -a], [a, hi[1] + a, a]), ([-a, lo[1] - a, -a], [a, lo[1], a])]
    return np.array([np.concatenate([np.asarray(x, float), np.asarray(y, float)]) for x, y in b])


FREEFLYER_OBSTACLE_CENTERS = np.array([
    [0.460, 0.315, 0.0], [0.201, 1.085, 0.0], [0.540, 2.020, 0.0], [1.374, 0.196, 0.0], [1.063, 1.354, 0.0],
    [1.365, 2.322, 0.0], [2.221, 0.548, 0.0], [2.077, 1.443, 0.0], [3.098, 1.186, 0.0], [2.837, 2.064, 0.0]])


def freeflyer_notebook_boxes():
    """The 10 inflated box obstacles of examples/freeflyerSE2.ipynb cell 2.  The notebook builds them as
    HyperRectangle(Vec3f0(...)), i.e. through Float32 -- reproduced by the float32 round trip."""
    w = np.array([0.27, 0.27, 0.127])
    infl = 0.05 * np.ones(3)
    out = []
    for c in FREEFLYER_OBSTACLE_CENTERS:
        mn = (c - 0.5 * w - infl + np.array([0.0, 0.0, 0.5 * w[0]])).astype(np.float32).astype(np.float64)
        sz = (w + 2 * infl).astype(np.float32).astype(np.float64)
        out.append(np.concatenate([mn, mn + sz]))
    return np.array(out)


def freeflyer_env():
    """keepout_zones then obstacle_set, the order Workspace(robot, env) uses (src/types.jl:19)."""
    return np.vstack([table_stanford_boxes(), freeflyer_notebook_boxes()])


FREEFLYER_X_INIT = np.array([0.2, 2.4, 0.0, 0.0, 0.0, 0.0])
FREEFLYER_X_GOAL = np.array([3.0, 0.5, 0.0, 0.05, -0.05, 0.0])
FREEFLYER_TF = 200.0
FREEFLYER_RADIUS = 0.157


def splitmix64(seed):
    """splitmix64 stream; doubles = (u >> 11) * 2^-53."""
    mask = (1 << 64) - 1
    state = seed & mask
    while True:
        state = (state + 0x9E3779B97F4A7C15) & mask
        z = state
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & mask
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & mask
        z = z ^ (z >> 31)
        yield (z >> 11) * (2.0 ** -53)


def _sdf_rect2(p, lo, hi):
    dx = max(lo[0] - p[0], 0.0, p[0] - hi[0])
    dy = max(lo[1] - p[1], 0.0, p[1] - hi[1])
    if dx > 0 or dy > 0:
        return float(np.hypot(dx, dy))
    return -min(p[0] - lo[0], hi[0] - p[0], p[1] - lo[1], hi[1] - p[1])


def freeflyer_random_x_init(B, first=0):
    """Config 2: position uniform over [0.25,3.40]x[0.25,2.49], rejected while the body is closer than 0.10 m to
    any box; theta = 0, v = omega = 0.  Problem b uses its own splitmix64 stream seeded 0x9E3779B97F4A7C15 + b."""
    env = freeflyer_env()
    X = np.zeros((B, 6))
    for i in range(B):
        g = splitmix64(0x9E3779B97F4A7C15 + first + i)
        while True:
            p = np.array([0.25 + (3.40 - 0.25) * next(g), 0.25 + (2.49 - 0.25) * next(g)])
            if min(_sdf_rect2(p, bx[0:2], bx[3:5]) for bx in env) - FREEFLYER_RADIUS >= 0.10:
                break
        X[i, 0:2] = p
    return X


def freeflyer_batch(B, first=0):
    """(x_init, goal_lo, goal_hi, tf) of config 2."""
    x0 = freeflyer_random_x_init(B, first)
    goal = np.tile(FREEFLYER_X_GOAL, (B, 1))
    return x0, goal.copy(), goal.copy(), np.full(B, FREEFLYER_TF)


def dubins_batch(B, first=0):
    """Config 3: x_init uniform over [-3,3]^2 x [-pi,pi], goal (0,0,0), tf = 10."""
    X = np.zeros((B, 3))
    for i in range(B):
        g = splitmix64(0x9E3779B97F4A7C15 + first + i)
        X[i] = [-3 + 6 * next(g), -3 + 6 * next(g), -np.pi + 2 * np.pi * next(g)]
    goal = np.zeros((B, 3))
    return X, goal.copy(), goal.copy(), np.full(B, 10.0)


# ---- ISS corner (src/environment/iss_corner.jl + iss_corner.mat, converted by tools/convert_iss_corner.py) ----
def _iss():
    import os
    return np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "data", "iss_corner.npz"))


def iss_corner_env(with_obstacles=True):
    """ISSCorner() keep-out AABBs (26), then add_obstacles! (4 boxes, 2 spheres): returns (boxes, spheres)."""
    d = _iss()
    if with_obstacles:
        return np.vstack([d["keepout"], d["rectangles"]]), d["spheres"].copy()
    return d["keepout"].copy(), np.zeros((0, 4))


ASTROBEE_RADIUS = np.sqrt(3.0) * 0.5 * 0.305


def _sdf_box3(p, lo, hi):
    e = np.maximum(np.maximum(lo - p, 0.0), p - hi)
    if (e > 0).any():
        return float(np.linalg.norm(e))
    return -float(min((p - lo).min(), (hi - p).min()))


_ISS_CACHE = {}


def _astrobee_free_points(count, seed, with_obstacles=True, margin=0.1):
    """Points uniform in keep-in boxes #4,#5,#8 (the corner module) shrunk by r+margin, rejected if the sphere
    touches any keep-out component.  The obstacle test runs over all components at
    once (the same arithmetic per component as _sdf_box3; the tables are read from disk once)."""
    key = bool(with_obstacles)
    if key not in _ISS_CACHE:
        d = _iss()
        boxes, sph = iss_corner_env(with_obstacles)
        _ISS_CACHE[key] = (d["keepin"][[3, 4, 7]].copy(), boxes[:, :3].copy(), boxes[:, 3:].copy(), sph.copy())
    zones, blo, bhi, sph = _ISS_CACHE[key]
    g = splitmix64(seed)
    pts = []
    while len(pts) < count:
        z = zones[int(next(g) * 3) % 3]
        lo, hi = z[:3] + ASTROBEE_RADIUS + margin, z[3:] - ASTROBEE_RADIUS - margin
        p = lo + (hi - lo) * np.array([next(g), next(g), next(g)])
        e = np.maximum(np.maximum(blo - p, 0.0), p - bhi)
        out = (e > 0).any(axis=1)
        d_out = np.sqrt((e * e).sum(axis=1))
        d_in = -np.minimum((p - blo).min(axis=1), (bhi - p).min(axis=1))
        ok = bool((np.where(out, d_out, d_in) - ASTROBEE_RADIUS > margin).all())
        if ok and len(sph):
            ok = bool((np.sqrt(((p - sph[:, :3]) ** 2).sum(axis=1)) - sph[:, 3] - ASTROBEE_RADIUS > margin).all())
        if ok:
            pts.append(p)
    return np.array(pts)


def astrobee_se3_batch(B, first=0, tf=70.0):
    """Config 4: start/goal positions in the corner module, MRP p=0 start, goal MRP uniform |p| <= 0.4, rest to rest."""
    x0, xg = np.zeros((B, 12)), np.zeros((B, 12))
    for i in range(B):
        pts = _astrobee_free_points(2, 0x9E3779B97F4A7C15 + first + i)
        x0[i, :3], xg[i, :3] = pts[0], pts[1]
        g = splitmix64(0xD1B54A32D192ED03 + first + i)
        while True:
            p = 0.8 * np.array([next(g), next(g), next(g)]) - 0.4
            if np.linalg.norm(p) <= 0.4:
                break
        xg[i, 6:9] = p
    return x0, xg.copy(), xg.copy(), np.full(B, tf)


def astrobee_manifold_batch(B, first=0, tf=40.0, eps=1e-4):
    """Config 5: q_init = (1,0,0,0), q_goal = normalise((1,a,b,c)), a,b,c ~ U(-0.5,0.5); point goals on r,v,w and
    a +-eps box on q as in examples/astrobeeSE3manifold.ipynb cell 1."""
    x0, glo, ghi = np.zeros((B, 13)), np.zeros((B, 13)), np.zeros((B, 13))
    for i in range(B):
        pts = _astrobee_free_points(2, 0x9E3779B97F4A7C15 + first + i)
        x0[i, :3] = pts[0]
        x0[i, 6] = 1.0
        g = splitmix64(0xD1B54A32D192ED03 + first + i)
        q = np.array([1.0, next(g) - 0.5, next(g) - 0.5, next(g) - 0.5])
        q /= np.linalg.norm(q)
        glo[i, :3] = ghi[i, :3] = pts[1]
        glo[i, 6:10], ghi[i, 6:10] = q - eps, q + eps
    return x0, glo, ghi, np.full(B, tf)


def astrobee_manifold_notebook(eps=1e-4):
    """The notebook's own manifold problem (examples/astrobeeSE3manifold.ipynb cell 1): the small corner maneuver r_init =
    (11.2, -0.8, 5.6) -> r_goal = (10.9, 3.0, 5.0), q_init = (1, 0, 0, 0) -> q_goal = normalise((1, 0.2, 0.3, 0.4)) as a +-eps
    BoxGoal, point goals on r, v, w, rest to rest, tf_guess = 10.  Returns (x_init, goal_lo, goal_hi, tf)."""
    x0, glo, ghi = np.zeros(13), np.zeros(13), np.zeros(13)
    x0[:3] = [11.2, -0.8, 5.6]
    x0[6] = 1.0
    q = np.array([1.0, 0.2, 0.3, 0.4])
    q /= np.linalg.norm(q)
    glo[:3] = ghi[:3] = [10.9, 3.0, 5.0]
    glo[6:10], ghi[6:10] = q - eps, q + eps
    return x0, glo, ghi, 10.0


def astrobee_manifold_batch_tf10(B, first=0):
    """SURVEY.md 8(d), config 5 "reported separately": the config-5 generator at the NOTEBOOK's horizon tf = 10 (the config itself
    runs tf = 40), problem 0 of the stream being the notebook's own problem.  At tf = 10 a share of the random start / goal pairs
    is out of reach of the acceleration limits: those subproblems are infeasible on both sides (device and oracle alike)."""
    x0, glo, ghi, tf = astrobee_manifold_batch(B, first, tf=10.0)
    if first == 0 and B > 0:
        x0[0], glo[0], ghi[0], tf[0] = astrobee_manifold_notebook()
    return x0, glo, ghi, tf


# ---- batches whose problems bring their own obstacle layouts (north_star: "random initial states / obstacle layouts") ----
def freeflyer_random_layouts(B, first=0, keep=0.6, sphere_prob=0.3):
    """One keep-out set per problem: the four table slabs (always), each of the ten notebook boxes kept with probability
    `keep`, and with probability `sphere_prob` one extra sphere (a disc for the planar model) of radius 0.10..0.20 m away
    from start corner and goal.  Problem b draws from its own splitmix64 stream (seed 0xA0761D6478BD642F + first + b).
    Returns (boxes_list, spheres_list): [n_box_b, 6] and [n_sph_b, 4] arrays."""
    slabs, obs = table_stanford_boxes(), freeflyer_notebook_boxes()
    boxes, spheres = [], []
    for i in range(B):
        g = splitmix64(0xA0761D6478BD642F + first + i)
        sel = [o for o in obs if next(g) < keep]
        boxes.append(np.vstack([slabs] + [o[None] for o in sel]) if sel else slabs.copy())
        if next(g) < sphere_prob:
            c = np.array([0.8 + 1.8 * next(g), 0.6 + 1.4 * next(g), 0.0])
            spheres.append(np.array([[c[0], c[1], c[2], 0.10 + 0.10 * next(g)]]))
        else:
            spheres.append(np.zeros((0, 4)))
    return boxes, spheres
