"""Pin the CPU oracle against the reference's own golden vectors.

CPU-only.  These are the gates that make every later GPU-vs-oracle comparison meaningful.
"""
import numpy as np
import pytest

from oracle import pcl_oracle as orc


def xyz1(a):
    out = np.ones((len(a), 4), np.float32)
    out[:, :3] = a[:, :3]
    return out


def test_correspondences_bunny_397(bunny, golden):
    # test/registration/test_registration_api.cpp:83-104
    src, tgt = xyz1(bunny["bun0"]), xyz1(bunny["bun4"])
    tree = orc.KdTree(tgt)
    q, m, d2 = tree.correspondences(src)
    gold = np.asarray(golden["correspondences_original"], np.int32)
    assert len(q) == 397
    assert np.array_equal(q, gold[:, 0]) and np.array_equal(q, np.arange(397))
    assert np.array_equal(m, gold[:, 1])
    # brute force agrees bit-for-bit with the kd-tree
    bi, bd = orc.knn_bruteforce(tgt, src, 1)
    assert np.array_equal(bi[:, 0], m) and np.array_equal(bd[:, 0], d2)


def test_reciprocal_correspondences_bunny_53(bunny, golden):
    # test/registration/test_registration_api.cpp:107-128
    src, tgt = xyz1(bunny["bun0"]), xyz1(bunny["bun4"])
    q, m, _ = orc.KdTree(tgt).reciprocal_correspondences(orc.KdTree(src), src, tgt)
    gold = np.asarray(golden["correspondences_reciprocal"], np.int32)
    assert np.array_equal(q, gold[:, 0]) and np.array_equal(m, gold[:, 1])


def test_kdtree_hand_points_k10(golden):
    # test/kdtree/test_kdtree.cpp:226-289 (the XY and rescaled variants exercise
    # PointRepresentation, restated here as a pre-scaling of the coordinates)
    g = golden["kdtree_hand"]
    pts = np.asarray(g["points"], np.float32)
    qry = np.asarray([g["query"]], np.float32)
    for name, scale in (("xyz", (1, 1, 1)), ("xy", (1, 1, 0)), ("rescaled_123", (1, 2, 3))):
        s = np.asarray(scale, np.float32)
        for knn in (lambda: orc.KdTree(pts * s).knn(qry * s, 10),
                    lambda: orc.knn_bruteforce(pts * s, qry * s, 10)):
            idx, d2 = knn()
            assert idx[0].tolist() == g[name]["indices"], name
            assert np.allclose(d2[0], g[name]["distances"], atol=g["dist_tol"])


def test_kdtree_vs_bruteforce_grid_k20():
    # test/kdtree/test_kdtree.cpp:161-208: 11^3 grid, k=20 -- stronger here: identical results,
    # on a lattice full of exact distance ties (lowest index must win).
    g = np.arange(-5, 6, dtype=np.float32) * np.float32(0.1)
    pts = np.stack(np.meshgrid(g, g, g, indexing="ij"), -1).reshape(-1, 3)
    rng = np.random.default_rng(7)
    qry = np.concatenate([pts[::17], rng.uniform(-0.6, 0.6, (200, 3)).astype(np.float32)])
    ti, td = orc.KdTree(pts).knn(qry, 20)
    bi, bd = orc.knn_bruteforce(pts, qry, 20)
    assert np.array_equal(ti, bi) and np.array_equal(td, bd)
    assert np.all(np.diff(td, axis=1) >= 0)


@pytest.mark.parametrize("k", [1, 8, 64, 512])
def test_kdtree_vs_bruteforce_random(k):
    # test/search/test_search.cpp:292-364 (k in {1,8,64,512}, 1200 random points)
    rng = np.random.default_rng(k)
    pts = rng.uniform(0, 1, (1200, 3)).astype(np.float32)
    pts[5] = np.nan  # non-finite target points are dropped (kdtree_flann.hpp:443-452)
    qry = rng.uniform(0, 1, (300, 3)).astype(np.float32)
    ti, td = orc.KdTree(pts).knn(qry, k)
    bi, bd = orc.knn_bruteforce(pts, qry, k)
    assert np.array_equal(ti, bi) and np.array_equal(td, bd)
    assert not np.any(ti == 5)
    for row in ti[:20]:
        assert len(set(row.tolist())) == k


def test_k_clamped_to_cloud_size():
    pts = np.random.default_rng(0).uniform(0, 1, (5, 3)).astype(np.float32)
    idx, d2 = orc.KdTree(pts).knn(pts[:2], 8)
    assert np.all(idx[:, :5] >= 0) and np.all(idx[:, 5:] == -1) and np.all(np.isinf(d2[:, 5:]))


def test_icp_bunny_golden(bunny, golden):
    # test/registration/test_registration.cpp:236-270
    g = golden["icp_bunny"]
    src, tgt = xyz1(bunny["bun0"]), xyz1(bunny["bun4"])
    r = orc.icp_align(orc.KdTree(tgt), tgt, src, mode=0,
                      max_iterations=g["max_iterations"],
                      transformation_epsilon=g["transformation_epsilon"],
                      max_correspondence_distance=g["max_correspondence_distance"])
    T = r["T"]
    assert np.all(np.abs(T[:3] - np.asarray(g["rows"])) <= np.asarray(g["tol"])), T
    assert T[3].tolist() == [0, 0, 0, 1]
    assert r["converged"] and r["iterations"] < 50
    # double-accumulated variant lands on the same answer
    r2 = orc.icp_align(orc.KdTree(tgt), tgt, src, mode=0, acc_double=1,
                       max_iterations=g["max_iterations"],
                       transformation_epsilon=g["transformation_epsilon"],
                       max_correspondence_distance=g["max_correspondence_distance"])
    assert np.abs(r2["T"] - T).max() < 1e-5 and r2["iterations"] == r["iterations"]


def test_icp_translation_recovery(bunny):
    # test/registration/test_registration.cpp:161-195: z + 0.2 recovered to 2e-3
    src = xyz1(bunny["bun0"])
    tgt = src.copy()
    tgt[:, 2] += np.float32(0.2)
    r = orc.icp_align(orc.KdTree(tgt), tgt, src, mode=0, max_iterations=50)
    T = r["T"]
    assert np.abs(T[:3, :3] - np.eye(3)).max() < 2e-3
    assert np.abs(T[:3, 3] - [0, 0, 0.2]).max() < 2e-3


def test_svd_known_answer(bunny, golden):
    # test/registration/test_registration_api.cpp:383-424: recover T_ref to 1e-6 (quaternion+t)
    w, x, y, z = np.asarray(golden["R_ref_quat_wxyz_unnormalized"], np.float64) / np.linalg.norm(
        golden["R_ref_quat_wxyz_unnormalized"])
    R = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                  [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                  [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])
    T_ref = np.eye(4)
    T_ref[:3, :3] = R
    T_ref[:3, 3] = golden["t_ref"]
    src = xyz1(bunny["bun4"])
    tgt = orc.transform_cloud(T_ref.astype(np.float32), src, order=1)
    for acc in (False, True):
        T = orc.umeyama(src, tgt, acc_double=acc)
        assert np.abs(T - T_ref).max() < 2e-6, (acc, np.abs(T - T_ref).max())


def test_lls_known_answer(golden):
    # test/registration/test_registration_api.cpp:469-518 (tolerance 1e-2: small-angle model)
    xs = np.arange(-5.0, 5.0001, 0.5, dtype=np.float32)
    X, Y = np.meshgrid(xs, xs, indexing="ij")
    x, y = X.ravel(), Y.ravel()
    z = np.float32(0.1) * x ** 2 + np.float32(0.2) * x * y - np.float32(0.3) * y + np.float32(1.0)
    n = np.stack([-0.2 * x - 0.2, 0.6 * y - 0.2, np.ones_like(x)], 1).astype(np.float32)
    n /= np.linalg.norm(n, axis=1, keepdims=True)
    src = np.stack([x, y, z, np.ones_like(x)], 1).astype(np.float32)
    G = np.asarray(golden["lls_ground_truth"], np.float32)
    tgt, tn = orc.transform_cloud(G, src, order=1, normals=n)
    T, sums, used = orc.lls_point_to_plane(src, tgt, tn)
    assert used == 441
    assert np.abs(T - G).max() < golden["lls_tol"]
    assert np.array_equal(orc.lls_solve(sums), T)


def test_normal_bun0_plane_golden(bunny, golden):
    # test/features/test_normal_estimation.cpp:96-124: whole bun0 as one neighbourhood
    g = golden["normal_bun0_plane"]
    cloud = xyz1(bunny["bun0"])
    cov, cen, cnt = orc.mean_and_covariance(cloud, np.arange(len(cloud)))
    assert cnt == 397
    nx, ny, nz, curv = orc.solve_plane_parameters(cov)
    assert np.allclose(np.abs([nx, ny, nz]), g["abs_n"], atol=g["tol"])
    assert abs(curv - g["curvature"]) < g["tol"]
    d = -(nx * cen[0] + ny * cen[1] + nz * cen[2])  # feature.hpp:58-60
    assert abs(abs(d) - g["abs_d"]) < g["tol"]


def test_normals_knn_translation_invariance(bunny):
    # test/features/test_normal_estimation.cpp:287-312 (k=15, shift (123,-45,98))
    cloud = xyz1(bunny["bun0"])
    n1, nan1 = orc.KdTree(cloud).normals(cloud, 15)
    shifted = cloud.copy()
    shifted[:, :3] += np.array([123, -45, 98], np.float32)
    n2, nan2 = orc.KdTree(shifted).normals(shifted, 15,
                                           viewpoint=(123.0, -45.0, 98.0))
    assert nan1 == 0 and nan2 == 0
    dots = np.abs(np.sum(n1[:, :3] * n2[:, :3], axis=1))
    assert np.all(dots >= 1 - 1e-4)
    assert np.allclose(np.linalg.norm(n1[:, :3], axis=1), 1, atol=1e-5)


def test_icp_with_normals_bunny(bunny):
    # test/registration/test_registration.cpp:272-318: NormalEstimation(k=10) + ICPWithNormals
    # converges with fitness < 1e-3.
    src, tgt = xyz1(bunny["bun0"]), xyz1(bunny["bun4"])
    tree = orc.KdTree(tgt)
    nrm, nan = tree.normals(tgt, 10)
    assert nan == 0
    r = orc.icp_align(tree, tgt, src, mode=1, tgt_normals=nrm, max_iterations=50,
                      transformation_epsilon=1e-8)
    out = orc.transform_cloud(r["T"], src, order=1)
    _, d2 = tree.knn(out, 1)
    assert r["converged"] and float(d2.mean()) < 1e-3


def test_voxelgrid_bunny_golden(bunny, golden):
    # test/filters/test_filters.cpp:566-596
    g = golden["voxelgrid_bun0"]
    cloud = xyz1(bunny["bun0"])
    out, ids = orc.voxelgrid(cloud, g["leaf"])
    assert len(out) == g["count"] and np.all(np.diff(ids) > 0)
    out, _ = orc.voxelgrid(cloud, g["leaf"], limits=(g["z_min"], g["z_max"]))
    assert len(out) == g["count_z"]
    assert np.allclose(out[0, :3], g["first_z"], atol=g["tol"])
    assert np.allclose(out[-1, :3], g["last_z"], atol=g["tol"])


def test_convergence_criteria_state_machine():
    # impl/default_convergence_criteria.hpp:49-140
    c = orc.new_convergence()
    L = orc.lib()
    import ctypes as C
    I = np.eye(4, dtype=np.float32).reshape(16)
    fp = I.ctypes.data_as(C.POINTER(C.c_float))
    c.max_iterations = 3
    c.translation_threshold = 0.0
    c.mse_threshold_relative = -1e300
    big = I.copy()
    big[3] = 0.5
    bp = big.ctypes.data_as(C.POINTER(C.c_float))
    assert L.orc_convergence_has_converged(C.byref(c), 1, bp, 1.0) == 0
    assert L.orc_convergence_has_converged(C.byref(c), 2, bp, 0.5) == 0
    assert L.orc_convergence_has_converged(C.byref(c), 3, bp, 0.25) == 1
    assert c.convergence_state == 1  # CONVERGENCE_CRITERIA_ITERATIONS
    # identity increment -> TRANSFORM
    c2 = orc.new_convergence()
    assert L.orc_convergence_has_converged(C.byref(c2), 1, fp, 1.0) == 1
    assert c2.convergence_state == 2
    # same MSE twice -> ABS_MSE
    c3 = orc.new_convergence()
    c3.translation_threshold = 0.0
    assert L.orc_convergence_has_converged(C.byref(c3), 1, bp, 0.3) == 0
    assert L.orc_convergence_has_converged(C.byref(c3), 2, bp, 0.3) == 1
    assert c3.convergence_state == 3


def test_rejectors_bunny_goldens(bunny, golden):
    # test/registration/test_registration_api.cpp:131-380 (Distance 97, MedianDistance 139 +
    # median 0.000465391, OneToOne 103, Trimmed 198) on the 397 original correspondences
    from oracle import rejectors as rej
    src, tgt = xyz1(bunny["bun0"]), xyz1(bunny["bun4"])
    q, m, d = orc.KdTree(tgt).correspondences(src)
    rq, rm, _ = rej.reject_distance(q, m, d, golden["rej_dist_max_dist"])
    assert np.array_equal(np.stack([rq, rm], 1), np.asarray(golden["correspondences_dist"]))
    rq, rm, _, med = rej.reject_median_distance(q, m, d, golden["rej_median_factor"])
    assert abs(med - golden["rej_median_distance"]) < 1e-4
    assert np.array_equal(np.stack([rq, rm], 1), np.asarray(golden["correspondences_median_dist"]))
    rq, rm, _ = rej.reject_one_to_one(q, m, d)
    assert np.array_equal(np.stack([rq, rm], 1), np.asarray(golden["correspondences_one_to_one"]))
    rq, rm, _ = rej.reject_trimmed(q, m, d, golden["rej_trimmed_overlap"])
    assert np.array_equal(np.stack([rq, rm], 1), np.asarray(golden["correspondences_trimmed"]))


def test_radius_search_sac_plane_golden():
    # test/kdtree/test_kdtree.cpp:292-330: radiusSearch(0.02) lists of every point of sac_plane_test.pcd
    import os
    from oracle import rejectors as rej
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "sac_plane_radius.npz"))
    cloud = z["cloud"]
    off, idx, d2 = rej.radius_search_bruteforce(cloud, cloud, float(z["radius"]))
    assert np.array_equal(off.astype(np.int64), z["offsets"])
    assert np.array_equal(idx, z["indices"])


def test_fitness_score_known_answer():
    # test/registration/test_registration.cpp:198-229: (0 + 0 + 0 + 0.25) / 4 = 0.0625
    src = np.asarray([(0, 0, 0), (0, 1, 0), (0, 0, 1), (10, 0, 0)], np.float32)
    tgt = np.asarray([(0, 0, 0), (0, 1, 0), (0, 0, 1), (10, 0, 0.5)], np.float32)
    tree = orc.KdTree(tgt)
    score, nr = tree.fitness_score(src, np.eye(4, dtype=np.float32), 1.0)
    assert abs(score - 0.0625) < 1e-4 and nr == 4
    score, nr = tree.fitness_score(src, np.eye(4, dtype=np.float32), 0.2)
    assert score == 0.0 and nr == 3
    score, nr = tree.fitness_score(src, np.eye(4, dtype=np.float32), -1.0)
    assert score == np.finfo(np.float64).max and nr == 0


def _quadric_with_normals():
    # the test surface of test/registration/test_registration_api.cpp:469-518 and :663-712
    xs = np.arange(-5.0, 5.0001, 0.5, dtype=np.float32)
    X, Y = np.meshgrid(xs, xs, indexing="ij")
    x, y = X.ravel(), Y.ravel()
    z = np.float32(0.1) * x ** 2 + np.float32(0.2) * x * y - np.float32(0.3) * y + np.float32(1.0)
    n = np.stack([-0.2 * x - 0.2, 0.6 * y - 0.2, np.ones_like(x)], 1).astype(np.float32)
    n /= np.linalg.norm(n, axis=1, keepdims=True)
    return np.stack([x, y, z, np.ones_like(x)], 1).astype(np.float32), n


def test_symmetric_lls_known_answer(golden):
    # test/registration/test_registration_api.cpp:663-712 (tolerance 1e-2, same ground truth as the LLS test)
    src, sn = _quadric_with_normals()
    G = np.asarray(golden["lls_ground_truth"], np.float32)
    tgt, tn = orc.transform_cloud(G, src, order=1, normals=sn)
    for acc in (False, True):
        T, sums, used = orc.lls_symmetric(src, sn, tgt, tn, acc_double=acc)
        assert used == 441
        assert np.abs(T - G).max() < 1e-2, (acc, np.abs(T - G).max())
        assert np.array_equal(orc.symmetric_solve(sums), T)
    # flipped target normals: enforce_same_direction recovers the same answer, without it the system changes
    T1, _, _ = orc.lls_symmetric(src, sn, tgt, -tn, enforce_same_direction=True)
    assert np.abs(T1 - G).max() < 1e-2
    T2, _, _ = orc.lls_symmetric(src, sn, tgt, -tn, enforce_same_direction=False)
    assert not np.abs(T2 - G).max() < 1e-2


def test_gicp_covariances_properties():
    # impl/gicp.hpp:70-147 -- no golden vector in the reference's tests; pinned by construction:
    # cov = U diag(1, 1, eps) U^T, u3 = normal of the neighbourhood
    rng = np.random.default_rng(9)
    n = 4000
    normal = np.array([0.3, -0.5, 0.81])
    normal /= np.linalg.norm(normal)
    a = np.cross(normal, [1, 0, 0])
    a /= np.linalg.norm(a)
    b = np.cross(normal, a)
    uv = rng.uniform(-1, 1, (n, 2))
    pts = (uv[:, :1] * a + uv[:, 1:] * b + 1e-5 * rng.normal(size=(n, 1)) * normal).astype(np.float32)
    tree = orc.KdTree(pts)
    cov = tree.gicp_covariances(pts, k=20, epsilon=0.001)
    w, v = np.linalg.eigh(cov)
    assert np.allclose(w, [0.001, 1.0, 1.0], atol=1e-12)
    assert np.abs(np.abs(v[:, :, 0] @ normal) - 1).max() < 1e-3          # smallest direction = plane normal
    assert np.allclose(cov, np.transpose(cov, (0, 2, 1)))
    with pytest.raises(ValueError):
        orc.KdTree(pts[:5]).gicp_covariances(pts[:5], k=20)


def test_oracle_normals_with_search_surface_is_consistent():
    # Feature::setSearchSurface restated (orc_normals_knn_queries): with the input as its own surface it is the plain
    # NormalEstimation; with other queries the plane comes from the SURFACE points and the flip from the QUERY point
    import pcl_amd
    from oracle import pcl_oracle as orc
    cloud = pcl_amd.synth.gaussian_surface(4000, pcl_amd.synth.TARGET_SEED)
    tree = orc.KdTree(cloud)
    a, na = tree.normals(cloud, 10, viewpoint=(0, 0, 10))
    b, nb = tree.normals_at(cloud, cloud, 10, viewpoint=(0, 0, 10))
    assert na == nb == 0 and np.array_equal(a, b)
    ind = np.arange(1, 4000, 13, dtype=np.int32)
    assert np.array_equal(tree.normals_at(cloud, cloud, 10, viewpoint=(0, 0, 10), indices=ind)[0], a[ind])
    q = cloud[:50].copy()
    q[:, 2] += np.float32(0.5)                       # queries above the sheet ...
    up, _ = tree.normals_at(cloud, q, 10, viewpoint=(0, 0, 10))
    down, _ = tree.normals_at(cloud, q, 10, viewpoint=(0, 0, -10))
    assert np.all(up[:, 2] > 0) and np.array_equal(down[:, :3], -up[:, :3])   # ... flipped as seen from the query
    idx, _ = tree.knn(q, 10)
    cov, cen, cnt = orc.mean_and_covariance(cloud, idx[0])
    n0 = np.array(orc.solve_plane_parameters(cov)[:3], np.float32)
    assert cnt == 10 and abs(abs(float(np.dot(n0, up[0, :3]))) - 1) < 1e-6      # the plane of the SURFACE neighbours
