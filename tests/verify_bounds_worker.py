"""Worker of tests/test_gpu_verify_bounds.py and tests/test_wavesim.py: runs in a subprocess whose PCLHIP_LIB points at a
-DPCLHIP_VERIFY_BOUNDS build of the library (scripts/build_variant.sh verify "-DPCLHIP_VERIFY_BOUNDS"; on the CPU tier
the emulation built with the same flag).  In that build every leaf a disc bound or a reach filter culls has its TRUE
minimum distance to the lane's query evaluated; slot 6 of the work counters counts the claims, slot 7 the broken ones.

  python tests/verify_bounds_worker.py <points> <family> [<family> ...]

Prints one JSON line per (family, scenario): {"family", "scenario", "checks", "violations", "matches_vs_oracle"}.
The launches are the ones that use the inexact bounds: the first launch of an alignment from several stand-offs (the
stand-off search where its gates pass, traverse() with discs otherwise), then seeded launches.
"""
import json
import sys

import numpy as np

sys.path.insert(0, ".")
import pcl_amd                              # noqa: E402
from pcl_amd import synth                   # noqa: E402


def rigid(rx=0.0, ry=0.0, rz=0.0, t=(0, 0, 0)):
    cx, sx, cy, sy, cz, sz = np.cos(rx), np.sin(rx), np.cos(ry), np.sin(ry), np.cos(rz), np.sin(rz)
    Rx = np.array([[1, 0, 0], [0, cx, -sx], [0, sx, cx]])
    Ry = np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]])
    Rz = np.array([[cz, -sz, 0], [sz, cz, 0], [0, 0, 1]])
    T = np.eye(4)
    T[:3, :3] = Rz @ Ry @ Rx
    T[:3, 3] = t
    return T.astype(np.float32)


def cloud(kind, n, seed):
    rng = np.random.default_rng(seed)
    if kind in synth.FAMILIES:
        return synth.family_cloud(kind, n, seed)
    out = np.ones((n, 4), np.float32)
    if kind == "collinear":                     # leaves whose points lie on lines: degenerate discs
        t = rng.uniform(0, 1, n)
        k = rng.integers(0, 40, n)
        out[:, 0] = t
        out[:, 1] = k * 0.025
        out[:, 2] = 0.3 * np.sin(6 * t)
    elif kind == "coincident":                  # many copies of few sites
        sites = rng.uniform(-1, 1, (max(n // 48, 1), 3))
        out[:, :3] = sites[rng.integers(0, len(sites), n)]
    elif kind == "far":                         # a unit scene 2500 units from the origin
        out[:, :3] = synth.family_cloud("sheet", n, seed)[:, :3] + np.float32(2500.0)
    elif kind == "mm":                          # a millimetre-sized scene
        out[:, :3] = synth.family_cloud("sheet", n, seed)[:, :3] * np.float32(1e-3)
    else:
        raise SystemExit("unknown family %r" % kind)
    return out


def main():
    n = int(sys.argv[1])
    with_oracle = n <= 300_000
    if with_oracle:
        from oracle import pcl_oracle as orc
    ctx = pcl_amd.Context(0)
    assert b"verify" in ctx.lib.pclhip_version(), "this worker needs a -DPCLHIP_VERIFY_BOUNDS build (PCLHIP_LIB)"
    for kind in sys.argv[2:]:
        tgt = cloud(kind, n, synth.TARGET_SEED)
        src = cloud(kind, n, synth.SOURCE_SEED)
        scale = float(np.abs(tgt[:, :3] - tgt[:, :3].mean(axis=0)).max())
        icp = pcl_amd.IterativeClosestPoint(ctx)
        icp.setInputTarget(tgt)
        icp.setInputSource(src)
        otree = orc.KdTree(tgt) if with_oracle else None
        stand = [("bench", np.linalg.inv(synth.ground_truth_transform()).astype(np.float32) if scale > 0.1 and scale < 10 else
                  rigid(rz=0.03, t=(0.01 * scale, -0.01 * scale, 0.015 * scale))),
                 ("lift", rigid(t=(0, 0, 0.05 * scale))), ("tilt", rigid(rx=0.08, ry=-0.05, t=(0.02 * scale, 0, 0.03 * scale))),
                 ("far", rigid(t=(0.3 * scale, 0.2 * scale, 0.6 * scale))), ("on", np.eye(4, dtype=np.float32))]
        for name, T in stand:
            icp.reset()
            ctx.counters(True)
            icp.iterate(T, max_dist=1e3 * scale)                 # the launch that starts an alignment
            ok = None
            if with_oracle:
                q, m, d = icp.fetchCorrespondences()
                oq, om, od = otree.correspondences(orc.transform_cloud(T, src, order=0), 1e3 * scale)
                ok = bool(np.array_equal(q, oq) and np.array_equal(m, om) and np.array_equal(d, od))
            icp.iterate(rigid(t=(1e-3 * scale, 0, 0)), max_dist=1e3 * scale)    # seeded
            icp.iterate(np.eye(4, dtype=np.float32), max_dist=0.02 * scale)      # seeded, with a maximum distance
            c = ctx.counters(False)
            print(json.dumps({"family": kind, "points": n, "scenario": name, "checks": c[6], "violations": c[7],
                              "matches_vs_oracle": ok}), flush=True)
        del icp


if __name__ == "__main__":
    main()
