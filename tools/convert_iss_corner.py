"""One-off converter: src/environment/iss_corner.mat of the reference -> static obstacle tables (.npz).
ISSCorner() builds HyperRectangle(Vec3f0(corner1), Vec3f0(corner2 - corner1)) (iss_corner.jl:11-24) and
add_obstacles! appends `rectangles` the same way and `spheres` as HyperSphere(Point3f0(c), Float32(r))
(iss_corner.jl:52-63): everything passes through Float32, reproduced here.  Run in the build container only
(needs /root/reference); the resulting gusto.jl_amd/data/iss_corner.npz is data, committed."""
import numpy as np
import scipy.io

m = scipy.io.loadmat("/root/reference/src/environment/iss_corner.mat", squeeze_me=True, struct_as_record=False)
f32 = lambda a: np.asarray(a, dtype=np.float64).astype(np.float32)


def boxes(zs):
    out = []
    for z in zs:
        c1 = f32(z.corner1)
        w = f32(np.asarray(z.corner2, float) - np.asarray(z.corner1, float))
        lo, hi = c1.astype(np.float64), (c1 + w).astype(np.float64)   # origin + widths, Float32 arithmetic
        out.append(np.concatenate([np.minimum(lo, hi), np.maximum(lo, hi)]))
    return np.array(out)


keepin = boxes(m["keepin_zones"])
keepout = boxes(m["keepout_zones"])
rects = boxes(m["rectangles"])
sph = np.array([np.concatenate([f32(s.center).astype(np.float64), [float(np.float32(s.radius))]]) for s in m["spheres"]])
np.savez("gusto.jl_amd/data/iss_corner.npz", keepin=keepin, keepout=keepout, rectangles=rects, spheres=sph)
print(keepin.shape, keepout.shape, rects.shape, sph.shape)
