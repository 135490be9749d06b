"""numpy restatement of PCL's correspondence rejectors (list -> list).  TEST INFRASTRUCTURE ONLY
(same rules as pcl_oracle.h).  Correspondences are (index_query int32[], index_match int32[],
distance float32[] = squared distance), citations relative to the PCL tree.

Where the reference uses an unstable std::sort the order of exact ties is unspecified; the
restatement (and the GPU) break such ties by the lower query index."""
import numpy as np


def reject_distance(q, m, d, max_distance):
    # registration/src/correspondence_rejection_distance.cpp:43-68; setMaximumDistance squares the
    # float (correspondence_rejection_distance.h:93-97), the test is `distance < max_distance_`
    md = np.float32(max_distance) * np.float32(max_distance)
    k = d < md
    return q[k], m[k], d[k]


def reject_median_distance(q, m, d, factor):
    # registration/src/correspondence_rejection_median_distance.cpp:43-69 (doubles, nth_element at n/2)
    dd = d.astype(np.float64)
    median = np.sort(dd)[len(dd) // 2]
    k = dd <= median * float(factor)
    return q[k], m[k], d[k], float(median)


def reject_one_to_one(q, m, d):
    # registration/src/correspondence_rejection_one_to_one.cpp:43-66: sort by (match, distance), keep
    # the first of every match index; output ordered by match index
    order = np.lexsort((q, d, m))
    q, m, d = q[order], m[order], d[order]
    first = np.ones(len(m), bool)
    first[1:] = m[1:] != m[:-1]
    first &= m >= 0
    return q[first], m[first], d[first]


def reject_trimmed(q, m, d, overlap_ratio, nr_min_correspondences=0):
    # registration/src/correspondence_rejection_trimmed.cpp:43-60
    n = int(np.floor(np.float32(overlap_ratio) * np.float32(len(d))))
    n = max(n, int(nr_min_correspondences))
    if n < len(d):
        order = np.lexsort((q, d))  # by distance, ties by query index
        q, m, d = q[order][:n], m[order][:n], d[order][:n]
    return q, m, d


def icp_with_filters(orc, tgt, src, mode, tgt_normals=None, rejectors=(), reciprocal=False, max_iterations=10,
                     max_correspondence_distance=None, transformation_epsilon=0.0, src_normals=None,
                     enforce_same_direction=True):
    """IterativeClosestPoint::computeTransformation (impl/icp.hpp:113-268) composed from the C oracle's
    pieces, with the rejector chain (:187-201) and reciprocal correspondences (:176-184).
    rejectors: list of callables (q, m, d) -> (q, m, d).  Returns dict like orc.icp_align."""
    import numpy as _np
    tree = orc.KdTree(tgt)
    order = 0 if mode == 0 else 1
    cur = _np.ascontiguousarray(src[:, :4], _np.float32).copy()
    cur_n = None if src_normals is None else _np.ascontiguousarray(src_normals, _np.float32).copy()
    final_T = _np.eye(4, dtype=_np.float32)
    conv = orc.new_convergence()
    conv.max_iterations = max_iterations
    conv.mse_threshold_relative = -_np.finfo(_np.float64).max
    conv.translation_threshold = transformation_epsilon
    md = max_correspondence_distance if max_correspondence_distance is not None else _np.sqrt(_np.finfo(_np.float64).max)
    it = 0
    import ctypes as C
    L = orc.lib()
    per_iter = []
    while True:
        if reciprocal:
            q, m, d = tree.reciprocal_correspondences(orc.KdTree(cur), cur, tgt, md)
        else:
            q, m, d = tree.correspondences(cur, md)
        for r in rejectors:
            out = r(q, m, d)
            q, m, d = out[0], out[1], out[2]
        per_iter.append((q.copy(), m.copy()))
        if len(q) < 3:
            return {"T": final_T, "iterations": it, "converged": False, "state": 5, "per_iter": per_iter}
        if mode == 1:
            Tk, _, _ = orc.lls_point_to_plane(cur, tgt, tgt_normals, q, m)
        elif mode == 2:  # setUseSymmetricObjective (icp.h:380-400): the source normals move with the cloud
            Tk, _, _ = orc.lls_symmetric(cur, cur_n, tgt, tgt_normals, q, m, enforce_same_direction, acc_double=True)
        else:
            Tk = orc.umeyama(cur, tgt, q, m, acc_double=True)
        if cur_n is not None:
            cur, cur_n = orc.transform_cloud(Tk, cur, order=order, normals=cur_n)
        else:
            cur = orc.transform_cloud(Tk, cur, order=order)
        final_T = orc.mat4_mul(Tk, final_T)
        it += 1
        mse = float(d.astype(_np.float64).sum() / len(d))
        Tf = _np.ascontiguousarray(Tk, _np.float32).reshape(16)
        if L.orc_convergence_has_converged(C.byref(conv), it, Tf.ctypes.data_as(C.POINTER(C.c_float)), mse):
            return {"T": final_T, "iterations": it, "converged": True, "state": conv.convergence_state, "per_iter": per_iter}
        if conv.convergence_state != 0:
            return {"T": final_T, "iterations": it, "converged": False, "state": conv.convergence_state, "per_iter": per_iter}


def radius_search_bruteforce(tgt, qry, radius, max_nn=0):
    """KdTreeFLANN::radiusSearch (kdtree/include/pcl/kdtree/impl/kdtree_flann.hpp:372-414), brute
    force in float32: d2 = ((dx*dx)+dy*dy)+dz*dz < float32(radius*radius), ascending (d2, index),
    optionally only the max_nn nearest.  Returns CSR (offsets, indices, d2)."""
    t = np.ascontiguousarray(tgt[:, :3], np.float32)
    r2 = np.float32(np.float64(radius) * np.float64(radius))
    offsets, idx, dd = [0], [], []
    fin = np.isfinite(t).all(1)
    for qp in np.ascontiguousarray(qry[:, :3], np.float32):
        if not np.isfinite(qp).all():
            offsets.append(offsets[-1])
            continue
        d = t - qp
        d = -d  # (q - c): same squares, kept explicit for clarity of the op order below
        d2 = (d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1]) + d[:, 2] * d[:, 2]
        sel = np.nonzero(fin & (d2 < r2))[0]
        order = np.lexsort((sel, d2[sel]))
        sel = sel[order]
        if max_nn and len(sel) > max_nn:
            sel = sel[:max_nn]
        idx.append(sel.astype(np.int32))
        dd.append(d2[sel])
        offsets.append(offsets[-1] + len(sel))
    return (np.asarray(offsets, np.uint64), np.concatenate(idx) if idx else np.zeros(0, np.int32),
            np.concatenate(dd) if dd else np.zeros(0, np.float32))


def normals_radius_at(orc, surface, queries, radius, viewpoint=(0.0, 0.0, 0.0)):
    """The same with a search surface different from the input (Feature::setSearchSurface): neighbours of every
    query among the SURFACE points, normal flipped as seen from the query.  Small clouds only."""
    surface = np.ascontiguousarray(surface, np.float32)
    queries = np.ascontiguousarray(queries, np.float32)
    off, idx, _ = radius_search_bruteforce(surface, queries, radius)
    out = np.full((len(queries), 4), np.nan, np.float32)
    nan = 0
    vp = np.asarray(viewpoint, np.float32)
    for i in range(len(queries)):
        nb = idx[int(off[i]):int(off[i + 1])]
        if not np.isfinite(queries[i, :3]).all() or len(nb) < 3:
            nan += 1
            continue
        cov, cen, cnt = orc.mean_and_covariance(surface, nb)
        if cnt == 0:
            nan += 1
            continue
        nx, ny, nz, curv = orc.solve_plane_parameters(cov)
        v = vp - queries[i, :3]
        cos_theta = np.float32(v[0] * np.float32(nx) + v[1] * np.float32(ny)) + v[2] * np.float32(nz)
        if cos_theta < 0:
            nx, ny, nz = -nx, -ny, -nz
        out[i] = (nx, ny, nz, curv)
    return out, nan


def normals_radius(orc, cloud, radius, viewpoint=(0.0, 0.0, 0.0)):
    """NormalEstimation with setRadiusSearch (Feature::compute, features/include/pcl/features/impl/feature.hpp
    :140-155 -> normal_3d.hpp:48-95): plane fit over all neighbours within the radius in the order
    radiusSearch returns them (ascending distance, ties by index); fewer than 3 -> NaN (normal_3d.h:308-322).
    Small clouds only (brute-force search, Python loop).  Returns ((n,4) float32, nan_count)."""
    cloud = np.ascontiguousarray(cloud, np.float32)
    n = len(cloud)
    off, idx, _ = radius_search_bruteforce(cloud, cloud, radius)
    out = np.full((n, 4), np.nan, np.float32)
    nan = 0
    vp = np.asarray(viewpoint, np.float32)
    for i in range(n):
        nb = idx[int(off[i]):int(off[i + 1])]
        if not np.isfinite(cloud[i, :3]).all() or len(nb) < 3:
            nan += 1
            continue
        cov, cen, cnt = orc.mean_and_covariance(cloud, nb)
        if cnt == 0:
            nan += 1
            continue
        nx, ny, nz, curv = orc.solve_plane_parameters(cov)
        v = vp - cloud[i, :3]                       # flipNormalTowardsViewpoint, normal_3d.h:169-188
        cos_theta = np.float32(v[0] * np.float32(nx) + v[1] * np.float32(ny)) + v[2] * np.float32(nz)
        if cos_theta < 0:
            nx, ny, nz = -nx, -ny, -nz
        out[i] = (nx, ny, nz, curv)
    return out, nan
