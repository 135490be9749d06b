#!/usr/bin/env python3
"""Model of the per-lane resolve phase of a seeded ICP iteration (numpy + scipy, CPU only; not a benchmark).

Question: with the previous match as seed, after evaluating the seed's 16-point leaf, for which fraction of the queries
does the ball (q, d) lie strictly inside the kd CELL of the seed's 16 / 64 / 256 / 1024-point block?  Those queries are
resolved exactly by looking at that block only.
"""
import sys
import time

import numpy as np
from scipy.spatial import cKDTree

sys.path.insert(0, ".")
from pcl_amd import synth  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
ITER = 5
t0 = time.time()
tgt, src, Tgt = synth.icp_pair(N)
tgt = tgt[:, :3].astype(np.float64)
src = src[:, :3].astype(np.float64)
print("clouds", time.time() - t0, flush=True)

# ---- kd order of the target: positional cuts at aligned boundaries, widest axis of the segment's box
LEAF = 16
nleaf = (N + LEAF - 1) // LEAF
R = 0
while 4 ** R < nleaf:
    R += 1
top = LEAF * 4 ** R
npad = top
perm = np.arange(N)
pts = tgt.copy()


def seg_boxes(p, S):
    """boxes of the aligned segments of S points (only real points); returns lo, hi [nseg,3], counts"""
    n = len(p)
    starts = np.arange(0, n, S)
    lo = np.minimum.reduceat(p, starts, axis=0)
    hi = np.maximum.reduceat(p, starts, axis=0)
    return lo, hi, starts


sizes = []  # (segment size, kind)
S = top
while S > 256:
    sizes.append((S, 4))
    S //= 4
while S > LEAF:
    sizes.append((S, 2))
    S //= 2
# cells: start with all space
cell_lo = np.full((1, 3), -np.inf)
cell_hi = np.full((1, 3), np.inf)
keep_cells = {}
for (S, way) in sizes:
    lo, hi, starts = seg_boxes(pts, S)
    nseg = len(starts)
    axis = np.argmax(hi - lo, axis=1)
    seg_id = np.arange(len(pts)) // S
    coord = pts[np.arange(len(pts)), axis[seg_id]]
    order = np.lexsort((coord, seg_id))
    pts = pts[order]
    perm = perm[order]
    # children cells
    C = S // way
    clo, chi, cstarts = seg_boxes(pts, C)
    nchild = len(cstarts)
    parent = (cstarts // S)
    new_lo = cell_lo[parent].copy()
    new_hi = cell_hi[parent].copy()
    ax = axis[parent]
    k = (cstarts % S) // C
    idx = np.arange(nchild)
    # lower face: max of the previous sibling along the axis; upper face: min of the next sibling
    has_prev = k > 0
    prev_hi = np.where(has_prev, chi[np.maximum(idx - 1, 0), ax], -np.inf)
    has_next = (k < way - 1) & (idx + 1 < nchild) & (((cstarts + C) // S) == parent)
    next_lo = np.where(has_next, clo[np.minimum(idx + 1, nchild - 1), ax], np.inf)
    new_lo[idx, ax] = np.maximum(new_lo[idx, ax], prev_hi)
    new_hi[idx, ax] = np.minimum(new_hi[idx, ax], next_lo)
    cell_lo, cell_hi = new_lo, new_hi
    if way == 4 and C > 256:
        pass
    keep_cells[C] = (cell_lo, cell_hi)
    # four-way cut = also the half cells (not needed)
print("kd order", time.time() - t0, "levels", [s for s, _ in sizes], flush=True)
leaf_lo, leaf_hi, _ = seg_boxes(pts, LEAF)

tree = cKDTree(pts, leafsize=16)
print("tree", time.time() - t0, flush=True)


def surf_normals(p):
    x, y = p[:, 0], p[:, 1]
    gx = np.zeros(len(p))
    gy = np.zeros(len(p))
    for cx, cy, s, a in synth._BUMPS:
        e = a * np.exp(-((x - cx) ** 2 + (y - cy) ** 2) / (2 * s * s))
        gx += e * (-(x - cx) / (s * s))
        gy += e * (-(y - cy) / (s * s))
    n = np.stack([-gx, -gy, np.ones(len(p))], axis=1)
    return n / np.linalg.norm(n, axis=1, keepdims=True)


nrm = surf_normals(pts)
q = src.copy()
prev = None
spacing = 2.0 / np.sqrt(N)
for it in range(1, ITER + 1):
    d, m = tree.query(q, k=1, workers=8)
    ok = d <= 0.1
    if prev is not None:
        # the seeded launch: seed = prev (position in kd order), queries q
        leaf = prev // LEAF
        P = pts.reshape(-1, LEAF, 3) if len(pts) % LEAF == 0 else None
        if P is None:
            padn = (-len(pts)) % LEAF
            P = np.concatenate([pts, np.full((padn, 3), 1e30)]).reshape(-1, LEAF, 3)
        # chunked leaf evaluation
        best = np.empty(len(q))
        for b in range(0, len(q), 1 << 20):
            e = min(len(q), b + (1 << 20))
            dd = ((P[leaf[b:e]] - q[b:e, None, :]) ** 2).sum(axis=2)
            best[b:e] = np.sqrt(dd.min(axis=1))
        seed_d = np.sqrt(((pts[prev] - q) ** 2).sum(axis=1))
        line = "it %d: seed dist %.2f spacings (median), after own leaf %.2f, true %.2f | seed leaf holds the answer %.3f |" % (
            it, np.median(seed_d) / spacing, np.median(best) / spacing, np.median(d) / spacing, np.mean(best <= d * (1 + 1e-12)))
        for C in (16, 32, 64, 128, 256, 1024, 4096):
            if C not in keep_cells:
                continue
            lo, hi = keep_cells[C]
            blk = prev // C
            inside = np.all((q - best[:, None] > lo[blk]) & (q + best[:, None] < hi[blk]), axis=1)
            line += " cell%d %.3f" % (C, inside.mean())
        print(line, flush=True)
        # how clustered are the unresolved (for cell 256)?
        for C in (64, 256):
            lo, hi = keep_cells[C]
            blk = prev // C
            inside = np.all((q - best[:, None] > lo[blk]) & (q + best[:, None] < hi[blk]), axis=1)
            un = np.flatnonzero(~inside)
            print("   cell%d: unresolved %d; span of 64 consecutive unresolved in source order (median, source points): %.0f" % (
                C, len(un), np.median(un[64::64] - un[:-64:64]) if len(un) > 128 else -1), flush=True)
    # point-to-plane step
    mm = m[ok]
    qq = q[ok]
    n = nrm[mm]
    r = ((pts[mm] - qq) * n).sum(axis=1)
    A = np.concatenate([np.cross(qq, n), n], axis=1)
    x = np.linalg.solve(A.T @ A, A.T @ r)
    al, be, ga = x[:3]
    Rm = np.array([[np.cos(ga) * np.cos(be), -np.sin(ga) * np.cos(al) + np.cos(ga) * np.sin(be) * np.sin(al), np.sin(ga) * np.sin(al) + np.cos(ga) * np.sin(be) * np.cos(al)],
                   [np.sin(ga) * np.cos(be), np.cos(ga) * np.cos(al) + np.sin(ga) * np.sin(be) * np.sin(al), -np.cos(ga) * np.sin(al) + np.sin(ga) * np.sin(be) * np.cos(al)],
                   [-np.sin(be), np.cos(be) * np.sin(al), np.cos(be) * np.cos(al)]])
    q = q @ Rm.T + x[3:]
    prev = m.copy()
    print("iteration", it, "mse", np.mean(d[ok] ** 2), "|x|", np.linalg.norm(x), time.time() - t0, flush=True)
