"""Deterministic synthetic clouds for the ICP hot path (SURVEY.md section 8(d) spec).

"Gaussian surface": eight asymmetric bumps on [-1,1]^2 (a single radial Gaussian is a surface of
revolution, which leaves yaw unobservable and makes the 6x6 normal system singular).  Points are
drawn with a counter-based RNG (splitmix64) so any shard [start, start+n) can be generated
independently by any rank -- this is what the multi-GPU bench uses.
"""
import numpy as np

_BUMPS = np.array([  # cx, cy, s, a
    (-.60, -.45, .22, .30), (.35, -.65, .18, -.20), (.70, .10, .25, .25), (-.15, .55, .30, .35),
    (-.75, .35, .15, -.15), (.10, -.10, .20, .20), (.55, .70, .17, -.25), (-.30, -.80, .12, .15),
], dtype=np.float64)

TARGET_SEED = 1001
SOURCE_SEED = 2002


def _splitmix64(x):
    x = (x + np.uint64(0x9E3779B97F4A7C15))
    z = x
    z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
    z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
    return z ^ (z >> np.uint64(31))


def _u01(seed, i, c):
    with np.errstate(over="ignore"):
        h = _splitmix64(np.uint64(seed) ^ _splitmix64(np.uint64(4) * i + np.uint64(c)))
    return (h >> np.uint64(40)).astype(np.float32) * np.float32(2.0 ** -24)


def surface_height(x, y):
    """S(x, y) in float64."""
    x = np.asarray(x, np.float64)
    y = np.asarray(y, np.float64)
    z = np.zeros_like(x)
    for cx, cy, s, a in _BUMPS:
        z += a * np.exp(-((x - cx) ** 2 + (y - cy) ** 2) / (2.0 * s * s))
    return z


def gaussian_surface(n, seed=TARGET_SEED, start=0, noise=1e-4, chunk=1 << 22):
    """(n, 4) float32 cloud x,y,z,1 -- points [start, start+n) of the stream for `seed`."""
    out = np.empty((n, 4), np.float32)
    for b in range(0, n, chunk):
        e = min(n, b + chunk)
        i = np.arange(start + b, start + e, dtype=np.uint64)
        x = np.float32(2.0) * _u01(seed, i, 0) - np.float32(1.0)
        y = np.float32(2.0) * _u01(seed, i, 1) - np.float32(1.0)
        u1 = _u01(seed, i, 2).astype(np.float64) + 2.0 ** -25
        u2 = _u01(seed, i, 3).astype(np.float64)
        g = np.sqrt(-2.0 * np.log(u1)) * np.cos(2.0 * np.pi * u2)
        z = surface_height(x, y) + noise * g
        out[b:e, 0] = x
        out[b:e, 1] = y
        out[b:e, 2] = z.astype(np.float32)
        out[b:e, 3] = 1.0
    return out


def ground_truth_transform():
    """T_gt: rotation 2 deg about (0.3,-0.5,0.81)/|.|, translation (0.012,-0.009,0.015); float64 4x4."""
    axis = np.array([0.3, -0.5, 0.81])
    axis /= np.linalg.norm(axis)
    th = np.deg2rad(2.0)
    K = np.array([[0, -axis[2], axis[1]], [axis[2], 0, -axis[0]], [-axis[1], axis[0], 0]])
    R = np.eye(3) + np.sin(th) * K + (1 - np.cos(th)) * (K @ K)
    T = np.eye(4)
    T[:3, :3] = R
    T[:3, 3] = (0.012, -0.009, 0.015)
    return T


def apply_rigid(T, pts):
    """float64 application of a 4x4 to an (n,>=3) float32 cloud; returns (n,4) float32, w=1."""
    T = np.asarray(T, np.float64)
    p = pts[:, :3].astype(np.float64)
    q = p @ T[:3, :3].T + T[:3, 3]
    out = np.empty((len(pts), 4), np.float32)
    out[:, :3] = q.astype(np.float32)
    out[:, 3] = 1.0
    return out


def icp_pair(n, start=0, n_target=None, target_start=0):
    """(target, source, T_gt): source = independent resample moved by T_gt^-1, so ICP returns ~T_gt."""
    tgt = gaussian_surface(n_target if n_target is not None else n, TARGET_SEED, target_start)
    src = gaussian_surface(n, SOURCE_SEED, start)
    T = ground_truth_transform()
    src = apply_rigid(np.linalg.inv(T), src)
    return tgt, src, T


# ---- the same stream generated where it is used: in device memory (bench.py --config 5: no rank materialises the
# 100M-point target on the host).  Same counter-based RNG bit for bit (int64 arithmetic wraps like uint64); the heights
# go through the device's exp / log / cos in double, so a z may differ from gaussian_surface()'s in its last float bit --
# the parity tests therefore keep the numpy generator, this one feeds measurements only.
def _lsr(z, k):
    return (z >> k) & ((1 << (64 - k)) - 1)


def _i64(v):
    v &= (1 << 64) - 1
    return v - (1 << 64) if v >= (1 << 63) else v


def _splitmix64_t(x):
    x = x + _i64(0x9E3779B97F4A7C15)
    z = (x ^ _lsr(x, 30)) * _i64(0xBF58476D1CE4E5B9)
    z = (z ^ _lsr(z, 27)) * _i64(0x94D049BB133111EB)
    return z ^ _lsr(z, 31)


def _u01_t(seed, i, c):
    import torch
    h = _splitmix64_t(_i64(int(seed)) ^ _splitmix64_t(4 * i + c))
    return _lsr(h, 40).to(torch.float32) * (2.0 ** -24)


def gaussian_surface_device(n, seed=TARGET_SEED, start=0, noise=1e-4, device="cuda", chunk=1 << 24):
    """gaussian_surface() as an (n, 4) float32 torch tensor generated on `device`."""
    import math
    import torch
    out = torch.empty((n, 4), dtype=torch.float32, device=device)
    for b in range(0, n, chunk):
        e = min(n, b + chunk)
        i = torch.arange(start + b, start + e, dtype=torch.int64, device=device)
        x = 2.0 * _u01_t(seed, i, 0) - 1.0
        y = 2.0 * _u01_t(seed, i, 1) - 1.0
        u1 = _u01_t(seed, i, 2).double() + 2.0 ** -25
        u2 = _u01_t(seed, i, 3).double()
        g = torch.sqrt(-2.0 * torch.log(u1)) * torch.cos(2.0 * math.pi * u2)
        xd, yd = x.double(), y.double()
        z = torch.zeros_like(xd)
        for cx, cy, s, a in _BUMPS:
            z += float(a) * torch.exp(-((xd - float(cx)) ** 2 + (yd - float(cy)) ** 2) / (2.0 * float(s) * float(s)))
        out[b:e, 0] = x
        out[b:e, 1] = y
        out[b:e, 2] = (z + noise * g).float()
        out[b:e, 3] = 1.0
    return out


def apply_rigid_device(T, pts):
    """apply_rigid() on a device tensor (float64 application, float32 result, w = 1)."""
    import torch
    Tt = torch.as_tensor(np.asarray(T, np.float64), device=pts.device)
    out = torch.empty((pts.shape[0], 4), dtype=torch.float32, device=pts.device)
    step = 1 << 24
    for b in range(0, pts.shape[0], step):
        p = pts[b:b + step, :3].double()
        out[b:b + step, :3] = (p @ Tt[:3, :3].T + Tt[:3, 3]).float()
    out[:, 3] = 1.0
    return out


# ---- cloud families beyond the bench's sheet (round 5): the geometry a search structure is NOT tuned on -----------------
# Same counter-based RNG, so every family is reproducible and shardable.  `kind`:
#   sheet     the Gaussian surface above (2.5-D)
#   cube      uniform in [-1, 1]^3 (volumetric, like the random clouds of test/search/test_search.cpp:292-364)
#   layers    two copies of the surface, the second lifted by three point spacings (a query between them has near
#             neighbours on both: leaf boxes of the two layers interleave in z)
#   clusters  half of the points spread over the surface, half packed into ten discs that cover 1 % of the area
#             (density contrast 100x: leaves of very different size next to each other)
FAMILIES = ("sheet", "cube", "layers", "clusters")
_CLUSTER_CENTRES = np.array([(-.7, -.6), (-.2, .7), (.5, .5), (.8, -.3), (0., 0.), (-.5, .1), (.3, -.7), (.65, .85), (-.85, .8),
                             (.1, .35)], np.float64)


def family_cloud(kind, n, seed=TARGET_SEED, start=0, noise=1e-4):
    """(n, 4) float32 cloud of family `kind` -- points [start, start+n) of the stream for `seed`."""
    if kind == "sheet":
        return gaussian_surface(n, seed, start, noise)
    i = np.arange(start, start + n, dtype=np.uint64)
    out = np.ones((n, 4), np.float32)
    if kind == "cube":
        for c in range(3):
            out[:, c] = np.float32(2.0) * _u01(seed, i, c) - np.float32(1.0)
        return out
    base = gaussian_surface(n, seed, start, noise)
    if kind == "layers":
        gap = 3.0 * 2.0 / np.sqrt(max(n, 2) / 2.0)
        upper = (i & np.uint64(1)).astype(bool)
        base[upper, 2] += np.float32(gap)
        return base
    if kind == "clusters":
        # every second point is re-drawn inside one of ten discs of radius r, 10 pi r^2 = 1 % of the 4 units of area
        r = np.sqrt(0.04 / (10.0 * np.pi))
        packed = (i & np.uint64(1)).astype(bool)
        which = ((i >> np.uint64(1)) % np.uint64(10)).astype(np.int64)
        u = _u01(seed ^ 0x5151, i, 0).astype(np.float64)
        v = _u01(seed ^ 0x5151, i, 1).astype(np.float64)
        rad = r * np.sqrt(u)
        x = _CLUSTER_CENTRES[which, 0] + rad * np.cos(2.0 * np.pi * v)
        y = _CLUSTER_CENTRES[which, 1] + rad * np.sin(2.0 * np.pi * v)
        u1 = _u01(seed ^ 0x5151, i, 2).astype(np.float64) + 2.0 ** -25
        u2 = _u01(seed ^ 0x5151, i, 3).astype(np.float64)
        g = np.sqrt(-2.0 * np.log(u1)) * np.cos(2.0 * np.pi * u2)
        z = surface_height(x, y) + noise * g
        base[packed, 0] = x[packed].astype(np.float32)
        base[packed, 1] = y[packed].astype(np.float32)
        base[packed, 2] = z[packed].astype(np.float32)
        return base
    raise ValueError("unknown cloud family %r" % (kind,))


def family_pair(kind, n, n_target=None):
    """(target, source, T_gt) of family `kind`: the source is an independent resample moved by T_gt^-1 (icp_pair's recipe)."""
    tgt = family_cloud(kind, n_target if n_target is not None else n, TARGET_SEED)
    src = family_cloud(kind, n, SOURCE_SEED)
    T = ground_truth_transform()
    return tgt, apply_rigid(np.linalg.inv(T), src), T
