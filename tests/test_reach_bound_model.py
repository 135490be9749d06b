"""CPU model test of the two disc bounds of pcl_amd/csrc/traverse.hpp (point_disc_lb, row_reach_alive) and of the disc
construction of index_build.hip (leaf_disc_kernel): numpy float32 mirrors of the device arithmetic, checked against brute
force on random geometry.  What must hold for the search to stay exact:
  * point_disc_lb(q, disc) <= the float32 L2_Simple distance of q to EVERY point of the leaf;
  * a (row, leaf) pair the reach filter drops is needed by no lane of the row (point_disc_lb > the lane's radius).
The GPU parity tests check the same end to end on the bench geometry; this one sweeps shapes, scales and stand-offs."""
import numpy as np
import pytest

F = np.float32


def l2_simple(q, p):  # FLANN L2_Simple in float32: ((dx*dx)+dy*dy)+dz*dz
    d = (q - p).astype(F)
    return (F(d[..., 0] * d[..., 0]) + F(d[..., 1] * d[..., 1])).astype(F) + F(d[..., 2] * d[..., 2])


def make_disc(pts):
    """leaf_disc_kernel: centre = rounded mean, direction of least variance shrunk by 4e-7, R and hn measured for them
    in double and rounded up."""
    p = pts.astype(np.float64)
    c = p.mean(0).astype(F)
    d = p - c.astype(np.float64)
    w, V = np.linalg.eigh(d.T @ d)
    n = (V[:, 0] / np.linalg.norm(V[:, 0]) * (1.0 - 4e-7)).astype(F)
    nn = float((n.astype(np.float64) ** 2).sum())
    assert 1.0 - 1e-6 <= nn <= 1.0
    R = np.nextafter(F(np.sqrt((d * d).sum(1).max()) * (1.0 + 1e-6)), F(np.inf))
    hn = np.nextafter(F(np.abs(d @ n.astype(np.float64)).max() * (1.0 + 1e-6)), F(np.inf))
    return c, R, n, hn


def point_disc_lb(q, c, R, n, hn):  # traverse.hpp: point_disc_lb (fma where the device uses fma)
    d = (q - c).astype(F)
    dx, dy, dz = d[..., 0], d[..., 1], d[..., 2]
    fma = lambda a, b, cc: (a.astype(np.float64) * b + cc).astype(F)
    r2 = fma(dz, dz, fma(dy, dy, F(dx * dx)))
    a = np.abs(fma(np.broadcast_to(n[2], dz.shape), dz, fma(np.broadcast_to(n[1], dy.shape), dy, F(n[0] * dx))))
    e = F(1e-6) * F(F(np.abs(dx) + np.abs(dy)) + np.abs(dz))
    a_hi = F(a + e)
    b2 = np.maximum(fma(r2, F(0.999999), -F(F(a_hi * a_hi) * F(1.000003))), F(0))
    gt = np.maximum(fma(np.sqrt(b2).astype(F), F(0.999999), -R), F(0))
    gn = np.maximum(F(F(a - e) - hn), F(0))
    return F(fma(gt, gt, F(gn * gn)) * F(0.999996))


def row_reach_alive(q, worst, ng, c, R, n, hn):  # traverse.hpp: the pair-mode block + row_reach_alive
    lo, hi = q.min(0), q.max(0)
    ctr = (F(0.5) * (lo + hi)).astype(F)
    ext = (hi - lo).astype(F)
    rS = F(np.sqrt(F(F(ext[0] * ext[0] + ext[1] * ext[1]) + ext[2] * ext[2])) * F(0.5000005)) + \
        F(1e-6) * F(F(abs(ctr[0]) + abs(ctr[1])) + abs(ctr[2]))
    a_own = (F(ng[0] * (q[:, 0] - ctr[0]) + ng[1] * (q[:, 1] - ctr[1])) + ng[2] * (q[:, 2] - ctr[2])).astype(F)
    rho_i = (np.sqrt(worst).astype(F) * F(1.000004)).astype(F)
    Up, Um, rho = (rho_i - a_own).max(), (rho_i + a_own).max(), rho_i.max()
    d = (ctr - c).astype(F)
    f64 = np.float64
    r2 = F(f64(d[2]) * d[2] + F(f64(d[1]) * d[1] + F(d[0] * d[0])))
    s0 = F(f64(n[2]) * d[2] + F(f64(n[1]) * d[1] + F(n[0] * d[0])))
    e = F(1e-6) * F(F(abs(d[0]) + abs(d[1])) + abs(d[2]))
    as0 = abs(s0)
    a_hi = F(as0 + e)
    b2 = max(F(f64(r2) * F(0.999999) - F(F(a_hi * a_hi) * F(1.000003))), F(0))
    gt = max(F(f64(np.sqrt(b2, dtype=F)) * F(0.999999) - F(rS + R)), F(0))
    alpha = min(max(F(f64(n[2]) * ng[2] + F(f64(n[1]) * ng[1] + F(n[0] * ng[0]))), F(-1)), F(1))
    m = np.array([F(f64(-alpha) * ng[k] + n[k]) for k in range(3)], F)
    mlen = F(np.sqrt(F(f64(m[2]) * m[2] + F(f64(m[1]) * m[1] + F(m[0] * m[0]))), dtype=F) * F(1.000001)) + F(2e-6)
    beta = -alpha if s0 < 0 else alpha
    ab = abs(beta)
    U = Um if beta < 0 else Up
    reach = F(F(f64(ab) * U + F(F(1 - ab) * rho)) - F(as0 - e)) + F(f64(mlen) * rS + hn)
    reach = F(reach + F(4e-6) * F(F(F(F(rho + as0) + F(rS + hn)) + abs(Up)) + abs(Um)))
    reach = min(reach, F(1e30))
    return bool(reach >= 0 and F(gt * gt) <= F(F(F(2.00002) * rho) * reach))


def _leaves(rng, kind, nleaf, scale, offset):
    if kind == "sheet":     # 4 x 4 patches of a sloped, noisy sheet
        gx, gy = np.meshgrid(np.arange(int(np.sqrt(nleaf))), np.arange(int(np.sqrt(nleaf))))
        out = []
        sx, sy = rng.uniform(-1, 1, 2)
        for a, b in zip(gx.ravel(), gy.ravel()):
            u = (a * 4 + rng.uniform(0, 4, 16)) * 6e-4
            v = (b * 4 + rng.uniform(0, 4, 16)) * 6e-4
            out.append(np.c_[u, v, sx * u + sy * v + 0.5 * u * v + 1e-4 * rng.normal(size=16)])
        L = np.array(out)
    elif kind == "blobs":   # volumetric clusters: discs are fat
        ctr = rng.uniform(-1, 1, (nleaf, 1, 3)) * 0.02
        L = ctr + rng.normal(size=(nleaf, 16, 3)) * 1.5e-3
    else:                    # needles: nearly collinear leaves
        ctr = rng.uniform(-1, 1, (nleaf, 1, 3)) * 0.02
        dirs = rng.normal(size=(nleaf, 1, 3))
        L = ctr + dirs * rng.uniform(-1, 1, (nleaf, 16, 1)) * 2e-3 + 1e-6 * rng.normal(size=(nleaf, 16, 3))
    return (L * scale + offset).astype(F)


@pytest.mark.parametrize("kind", ["sheet", "blobs", "needles"])
@pytest.mark.parametrize("scale,offset", [(1.0, 0.0), (300.0, 0.0), (1.0, 2500.0), (0.003, -7.0)])
def test_disc_bounds_never_exclude_a_needed_leaf(kind, scale, offset):
    import zlib
    rng = np.random.default_rng(zlib.crc32(repr((kind, scale, offset)).encode()))
    leaves = _leaves(rng, kind, 36, scale, offset)
    discs = [make_disc(l) for l in leaves]
    allpts = leaves.reshape(-1, 3)
    span = float(np.ptp(allpts, axis=0).max())
    checked = dropped = 0
    for trial in range(60):
        # a row of 16 queries: a small patch somewhere near / above / far from the leaves
        base = allpts[rng.integers(len(allpts))].astype(np.float64)
        standoff = span * float(10 ** rng.uniform(-3, 0.3)) * (rng.random() < 0.8)
        direction = rng.normal(size=3)
        direction /= np.linalg.norm(direction)
        patch = rng.normal(size=(16, 3)) * span * float(10 ** rng.uniform(-3, -1))
        q = (base + standoff * direction + patch).astype(F)
        # exact float32 distances to every point, per leaf
        dist = np.stack([l2_simple(q[:, None, :], l[None, :, :]).min(1) for l in leaves], 1)  # [16, nleaf]
        nn = dist.min(1)
        # the lanes' radii: the true nearest distance, inflated for some rows (bounds that have not tightened yet)
        worst = (nn * F(rng.choice([1.0, 1.0, 1.05, 2.0, 30.0]))).astype(F)
        ng = rng.normal(size=3) if rng.random() < 0.3 else np.linalg.svd(q - q.mean(0))[2][2]
        ng = (ng / np.linalg.norm(ng)).astype(F) if rng.random() < 0.9 else np.zeros(3, F)
        for j, (c, R, n, hn) in enumerate(discs):
            lb = point_disc_lb(q, c, R, n, hn)
            assert np.all(lb <= dist[:, j]), (kind, trial, j, float((lb - dist[:, j]).max()))
            needed = bool(np.any(lb <= worst))
            alive = row_reach_alive(q, worst, ng, c, R, n, hn)
            checked += 1
            dropped += not alive
            assert alive or not needed, (kind, scale, offset, trial, j)
    assert checked == 60 * 36 and dropped > 0   # the filter does filter on this geometry
