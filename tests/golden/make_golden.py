#!/usr/bin/env python3
"""Extract the reference's OWN golden vectors for the ICP hot path into small fixtures.

Run once in the build container (needs /root/reference, which does NOT exist on the GPU box):

    python tests/golden/make_golden.py

Inputs read (data and expected values only -- no reference source code is copied):
  test/bun0.pcd, test/bun4.pcd                                   (ASCII PCD fixtures)
  test/registration/test_registration_api_data.h:3-1124          (golden index pairs)
  test/registration/test_registration.cpp:251-269                (ICP 4x4 golden, 1e-3)
  test/kdtree/test_kdtree.cpp:228-282                            (10 hand points, k=10 orders)
  test/features/test_normal_estimation.cpp:105-124               (bun0 plane fit golden)
  test/filters/test_filters.cpp:576-596                          (VoxelGrid counts/centroids)
  test/sac_plane_test.pcd + test/kdtree/kdtree_unit_test_results.xml (radiusSearch 0.02 lists)
Outputs: tests/golden/bunny.npz, tests/golden/golden.json, tests/golden/sac_plane_radius.npz
"""
import json
import os
import re
import sys

import numpy as np

REF = os.environ.get("PCL_REFERENCE", "/root/reference")
HERE = os.path.dirname(os.path.abspath(__file__))


def read_pcd_ascii(path):
    """Minimal ASCII PCD reader (io/src/pcd_io.cpp:115-392 header, :456-559 body)."""
    fields, npts, data = None, None, []
    with open(path) as f:
        in_data = False
        for line in f:
            line = line.strip()
            if not line or line.startswith("#"):
                continue
            if in_data:
                data.append([float(v) for v in line.split()])
                continue
            key, *vals = line.split()
            if key == "FIELDS":
                fields = vals
            elif key == "POINTS":
                npts = int(vals[0])
            elif key == "DATA":
                assert vals[0] == "ascii"
                in_data = True
    arr = np.asarray(data, dtype=np.float32)
    assert arr.shape == (npts, len(fields)), (arr.shape, npts, fields)
    return fields, arr


def parse_pairs(text, name):
    m = re.search(r"const int %s\[(\d+)\]\[2\] = \{(.*?)\};" % name, text, re.S)
    n = int(m.group(1))
    nums = [int(v) for v in re.findall(r"-?\d+", m.group(2))]
    assert len(nums) == 2 * n, (name, len(nums), n)
    return np.asarray(nums, dtype=np.int32).reshape(n, 2).tolist()


def main():
    if not os.path.isdir(REF):
        sys.exit("reference tree not found at %s" % REF)
    f0, bun0 = read_pcd_ascii(os.path.join(REF, "test/bun0.pcd"))
    f4, bun4 = read_pcd_ascii(os.path.join(REF, "test/bun4.pcd"))
    assert f0[:3] == ["x", "y", "z"] and f4 == ["x", "y", "z"]
    np.savez_compressed(os.path.join(HERE, "bunny.npz"), bun0=bun0, bun4=bun4,
                        bun0_fields=np.asarray(f0))

    hdr = open(os.path.join(REF, "test/registration/test_registration_api_data.h")).read()
    gold = {
        "source": "PointCloudLibrary/pcl test tree (see make_golden.py docstring)",
        "correspondences_original": parse_pairs(hdr, "correspondences_original"),
        "correspondences_reciprocal": parse_pairs(hdr, "correspondences_reciprocal"),
        "correspondences_dist": parse_pairs(hdr, "correspondences_dist"),
        "rej_dist_max_dist": 0.01,
        "correspondences_median_dist": parse_pairs(hdr, "correspondences_median_dist"),
        "rej_median_factor": 0.5,
        "rej_median_distance": 0.000465391,
        "correspondences_one_to_one": parse_pairs(hdr, "correspondences_one_to_one"),
        "correspondences_trimmed": parse_pairs(hdr, "correspondences_trimmed"),
        "rej_trimmed_overlap": 0.5,
        # test_registration_api_data.h:1122-1123
        "R_ref_quat_wxyz_unnormalized": [0.9, 0.1, -0.25, 0.15],
        "t_ref": [0.5, -2.0, 1.0],
        # test/registration/test_registration.cpp:238-269 (bun0 -> bun4, 50 it, teps 1e-8, dmax .05)
        "icp_bunny": {
            "max_iterations": 50, "transformation_epsilon": 1e-8,
            "max_correspondence_distance": 0.05,
            "rows": [[0.8806, 0.036481287330389023, -0.4724, 0.03453],
                     [-0.02354, 0.9992, 0.03326, -0.001519],
                     [0.4732, -0.01817, 0.8808, 0.04116]],
            "tol": [[1e-3, 1e-2, 1e-3, 1e-3], [1e-3] * 4, [1e-3] * 4],
        },
        # test/kdtree/test_kdtree.cpp:228-282
        "kdtree_hand": {
            "points": [[86.6, 42.1, 92.4], [63.1, 18.4, 22.3], [35.5, 72.5, 37.3],
                       [99.7, 37.0, 8.7], [22.4, 84.1, 64.0], [65.2, 73.4, 18.0],
                       [60.4, 57.1, 4.5], [38.7, 17.6, 72.3], [14.2, 95.7, 34.7],
                       [2.5, 26.5, 66.0]],
            "query": [50.0, 50.0, 50.0],
            "xyz": {"indices": [2, 7, 5, 1, 4, 6, 9, 0, 8, 3],
                    "distances": [877.8, 1674.7, 1802.6, 1937.5, 2120.6, 2228.8, 3064.5,
                                  3199.7, 3604.2, 4344.8]},
            "xy": {"indices": [6, 2, 5, 1, 7, 0, 4, 3, 9, 8],
                   "distances": [158.6, 716.5, 778.6, 1170.2, 1177.5, 1402.0, 1924.6,
                                 2639.1, 2808.5, 3370.1]},
            "rescaled_123": {"indices": [2, 9, 4, 7, 1, 5, 8, 0, 3, 6],
                             "distances": [3686.9, 6769.2, 7177.0, 8802.3, 11071.5, 11637.3,
                                           11742.4, 17769.0, 18497.3, 18942.0]},
            "dist_tol": 0.1,
        },
        # test/features/test_normal_estimation.cpp:105-124 (whole bun0 as one neighbourhood)
        "normal_bun0_plane": {"abs_n": [0.035592, 0.369596, 0.928511], "abs_d": 0.0622552,
                              "curvature": 0.0693136, "tol": 1e-4,
                              "signed_n": [0.035592, 0.369596, 0.928511], "signed_d": -0.0622552},
        # test/filters/test_filters.cpp:572-596
        "voxelgrid_bun0": {"leaf": 0.02, "count": 103, "z_min": 0.05, "z_max": 0.1,
                           "count_z": 14, "first_z": [-0.026125, 0.039788, 0.052827],
                           "last_z": [-0.073202, 0.1296, 0.051333], "tol": 1e-4},
        # test/registration/test_registration_api.cpp:469-518 (LLS known answer, tol 1e-2)
        "lls_ground_truth": [[0.9938, 0.0988, 0.0517, 0.1], [-0.0997, 0.9949, 0.0149, -0.2],
                             [-0.05, -0.02, 0.9986, 0.3], [0, 0, 0, 1]],
        "lls_tol": 1e-2,
    }
    # test/kdtree/test_kdtree.cpp:292-330 + test/kdtree/kdtree_unit_test_results.xml: radiusSearch(0.02)
    # neighbour lists of every point of test/sac_plane_test.pcd
    fs, sac = read_pcd_ascii(os.path.join(REF, "test/sac_plane_test.pcd"))
    xml = open(os.path.join(REF, "test/kdtree/kdtree_unit_test_results.xml")).read()
    blocks = re.findall(r"<point_(\d+)>(.*?)</point_\1>", xml, re.S)
    assert len(blocks) == len(sac)
    offsets, flat = [0], []
    for i, (pid, body) in enumerate(blocks):
        assert int(pid) == i
        size = int(re.search(r"<size>(\d+)</size>", body).group(1))
        nn = [int(v) for v in re.findall(r"<nn_\d+>(\d+)</nn_\d+>", body)]
        assert len(nn) == size
        flat.extend(nn)
        offsets.append(len(flat))
    np.savez_compressed(os.path.join(HERE, "sac_plane_radius.npz"), cloud=sac[:, :3].astype(np.float32),
                        offsets=np.asarray(offsets, np.int64), indices=np.asarray(flat, np.int32),
                        radius=np.float64(0.02))

    # PCD fixtures: files the reference ships for its own tests (test/*.pcd), one per on-disk flavour, plus
    # the facts about them that pin the reader: POINTS, field list, extents of x/y/z after decoding with
    # oracle/pcd.py (whose LZF decoder must reproduce the stored uncompressed size exactly)
    import shutil
    sys.path.insert(0, os.path.join(HERE, "..", ".."))
    from oracle import pcd as opcd
    os.makedirs(os.path.join(HERE, "pcd"), exist_ok=True)
    facts = {}
    for name in ("curve_close.pcd",            # binary, x y z
                 "colored_cloud.pcd",          # binary, x y z rgb normal_x normal_y normal_z curvature
                 "ism_test.pcd",               # binary_compressed, x y z
                 "noisy_slice_displaced.pcd",  # binary_compressed, x y z
                 "bun0.pcd"):                  # ascii, x y z normal_x normal_y normal_z curvature
        src_path = os.path.join(REF, "test", name)
        shutil.copyfile(src_path, os.path.join(HERE, "pcd", name))
        h, fld, dense = opcd.read(src_path)
        facts[name] = {"data": h["data"], "points": h["points"], "width": h["width"], "height": h["height"],
                       "fields": [fl[0] for fl in h["fields"]], "is_dense": bool(dense),
                       "min": [float(fld[a].min()) for a in "xyz"], "max": [float(fld[a].max()) for a in "xyz"],
                       "bytes": os.path.getsize(src_path)}
    gold["pcd_files"] = facts
    # the ascii decoder agrees with the independent parser used for bunny.npz above
    assert np.array_equal(opcd.xyz(os.path.join(REF, "test/bun0.pcd"))[0][:, :3], bun0[:, :3].astype(np.float32))

    with open(os.path.join(HERE, "golden.json"), "w") as f:
        json.dump(gold, f, indent=0, separators=(",", ":"))
    print("wrote bunny.npz (%d + %d pts) and golden.json" % (len(bun0), len(bun4)))


if __name__ == "__main__":
    main()
