"""PCD reader / writer (host code of libpclhip.so, no GPU needed) against the oracle (oracle/pcd.py) and
the reference's own binary / binary_compressed / ascii test files (tests/golden/pcd/)."""
import json
import os

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
PCD = os.path.join(HERE, "golden", "pcd")


@pytest.fixture(scope="module")
def facts():
    return json.load(open(os.path.join(HERE, "golden", "golden.json")))["pcd_files"]


@pytest.fixture(scope="module")
def api():
    import pcl_amd
    return pcl_amd


def test_oracle_pinned_on_reference_files(facts):
    # the oracle's decoders reproduce the recorded facts (and, for binary_compressed, the stored
    # uncompressed size exactly -- asserted inside oracle.pcd.read)
    from oracle import pcd as opcd
    bunny = np.load(os.path.join(HERE, "golden", "bunny.npz"))
    for name, f in facts.items():
        h, fld, dense = opcd.read(os.path.join(PCD, name))
        assert os.path.getsize(os.path.join(PCD, name)) == f["bytes"]
        assert (h["data"], h["points"], h["width"], h["height"]) == (f["data"], f["points"], f["width"], f["height"])
        assert [x[0] for x in h["fields"]] == f["fields"] and dense == f["is_dense"]
        assert [float(fld[a].min()) for a in "xyz"] == f["min"] and [float(fld[a].max()) for a in "xyz"] == f["max"]
    # ascii: same numbers as the independent parser behind bunny.npz
    assert np.array_equal(opcd.xyz(os.path.join(PCD, "bun0.pcd"))[0][:, :3], bunny["bun0"][:, :3].astype(np.float32))


@pytest.mark.parametrize("name", ["curve_close.pcd", "colored_cloud.pcd", "ism_test.pcd", "noisy_slice_displaced.pcd",
                                  "bun0.pcd"])
def test_reader_matches_oracle_bit_exact(api, facts, name):
    from oracle import pcd as opcd
    path = os.path.join(PCD, name)
    info = api.getPCDHeader(path)
    f = facts[name]
    assert (info.points, info.width, info.height) == (f["points"], f["width"], f["height"])
    assert info.data_type == {"ascii": 0, "binary": 1, "binary_compressed": 2}[f["data"]]
    assert info.num_fields == len(f["fields"]) and info.has_xyz == 1
    assert info.has_normals == int("normal_x" in f["fields"]) and info.has_curvature == int("curvature" in f["fields"])
    assert info.has_rgb == int("rgb" in f["fields"] or "rgba" in f["fields"])
    assert info.version == 7 and list(info.viewpoint) == [0, 0, 0, 1, 0, 0, 0]
    for with_normals in (False, True):
        got, dense = api.loadPCDFile(path, with_normals=with_normals)
        want, wdense = opcd.xyz(path, with_normals=with_normals)
        assert dense == wdense == f["is_dense"]
        assert got.shape == want.shape and np.array_equal(got.view(np.uint32), want.view(np.uint32))


@pytest.mark.parametrize("mode", ["ascii", "binary", "binary_compressed"])
@pytest.mark.parametrize("with_normals", [False, True])
def test_write_read_round_trip(api, tmp_path, mode, with_normals):
    # test/io/test_io.cpp (PCDReaderWriter family): what is written is read back; binary modes bit-exact,
    # ascii to the 8 significant digits the reference writes by default
    from oracle import pcd as opcd
    rng = np.random.default_rng(5)
    n = 3000
    cloud = np.zeros((n, 12 if with_normals else 4), np.float32)
    cloud[:, :3] = rng.normal(0, 10, (n, 3))
    cloud[::7, :3] = np.round(cloud[::7, :3])   # runs of equal bytes for the compressor
    cloud[:, 3] = 1
    if with_normals:
        v = rng.normal(0, 1, (n, 3))
        cloud[:, 4:7] = v / np.linalg.norm(v, axis=1, keepdims=True)
        cloud[:, 8] = rng.uniform(0, 0.3, n)
    cloud[5, 0] = np.nan                         # "nan" in ascii, is_dense = false
    path = str(tmp_path / ("t_%s_%d.pcd" % (mode, with_normals)))
    api.savePCDFile(path, cloud, mode)
    info = api.getPCDHeader(path)
    assert (info.points, info.width, info.height, info.version) == (n, n, 1, 7)
    assert info.data_type == {"ascii": 0, "binary": 1, "binary_compressed": 2}[mode]
    assert info.has_normals == int(with_normals) and info.has_curvature == int(with_normals)
    back, dense = api.loadPCDFile(path, with_normals=with_normals)
    oback, odense = opcd.xyz(path, with_normals=with_normals)       # the oracle reads our files too
    assert not dense and not odense
    assert np.array_equal(back.view(np.uint32), oback.view(np.uint32))
    if mode == "ascii":
        assert np.allclose(back, cloud, rtol=1e-7, atol=0, equal_nan=True)
    else:
        assert np.array_equal(back.view(np.uint32), cloud.view(np.uint32))
    if mode == "binary_compressed":
        # random mantissas barely compress: LZF's worst case is one control byte per 32 literals
        assert os.path.getsize(path) < n * (28 if with_normals else 12) * 33 // 32 + 400


def test_lzf_edge_cases(api, tmp_path):
    # highly repetitive data (long back references, length extension byte) and incompressible data
    from oracle import pcd as opcd
    for k, cloud in enumerate((np.zeros((70000, 4), np.float32),
                               np.random.default_rng(1).integers(0, 2 ** 32, (5000, 4), dtype=np.uint32).view(np.float32),
                               np.ones((1, 4), np.float32), np.zeros((0, 4), np.float32))):
        cloud = np.ascontiguousarray(cloud)
        cloud = np.where(np.isfinite(cloud), cloud, np.float32(1.5)).astype(np.float32)
        path = str(tmp_path / ("e%d.pcd" % k))
        api.savePCDFile(path, cloud, "binary_compressed")
        back, _ = api.loadPCDFile(path)
        want = cloud.copy()
        want[:, 3] = 1
        assert np.array_equal(back.view(np.uint32), want.view(np.uint32))
        if len(cloud):
            assert np.array_equal(opcd.xyz(path)[0].view(np.uint32), want.view(np.uint32))


def test_reader_errors(api, tmp_path):
    p = tmp_path / "bad.pcd"
    p.write_text("VERSION 0.7\nFIELDS x y z\nSIZE 4 4\nTYPE F F F\nCOUNT 1 1 1\nWIDTH 1\nHEIGHT 1\nPOINTS 1\nDATA ascii\n0 0 0\n")
    with pytest.raises(api.PclHipError, match="SIZE"):
        api.getPCDHeader(str(p))
    p.write_text("VERSION 0.7\nFIELDS x y z\nSIZE 4 4 4\nTYPE F F F\nCOUNT 1 1 1\nWIDTH 2\nHEIGHT 2\nPOINTS 3\nDATA ascii\n")
    with pytest.raises(api.PclHipError, match="number of points"):
        api.getPCDHeader(str(p))
    p.write_text("VERSION 0.7\nFIELDS a b\nSIZE 4 4\nTYPE F F\nCOUNT 1 1\nWIDTH 1\nHEIGHT 1\nPOINTS 1\nDATA ascii\n1 2\n")
    with pytest.raises(api.PclHipError, match="x/y/z"):
        api.loadPCDFile(str(p))
    with pytest.raises(api.PclHipError, match="cannot open"):
        api.getPCDHeader(str(tmp_path / "missing.pcd"))
    # older header without SIZE/TYPE/COUNT: everything float32 (pcd_io.cpp:181-189); float64 fields convert
    p.write_text("FIELDS x y z\nWIDTH 2\nHEIGHT 1\nPOINTS 2\nDATA ascii\n1 2 3\n4 5 nan\n")
    c, dense = api.loadPCDFile(str(p))
    assert not dense and c[0].tolist() == [1, 2, 3, 1] and np.isnan(c[1, 2])
    assert api.getPCDHeader(str(p)).version == 6
    p.write_text("VERSION 0.7\nFIELDS x y z\nSIZE 8 8 8\nTYPE F F F\nCOUNT 1 1 1\nWIDTH 1\nHEIGHT 1\nVIEWPOINT 1 2 3 1 0 0 0\n"
                 "POINTS 1\nDATA ascii\n0.1 0.2 0.3\n")
    c, dense = api.loadPCDFile(str(p))
    assert dense and c[0].tolist() == [np.float32(0.1), np.float32(0.2), np.float32(0.3), 1.0]
    assert list(api.getPCDHeader(str(p)).viewpoint) == [1, 2, 3, 1, 0, 0, 0]


def test_reader_rejects_hostile_headers(api, tmp_path):
    # untrusted files: negative / odd SIZE, bad TYPE, a second FIELDS line with more fields than SIZE/TYPE,
    # field extents beyond the record, POINTS that the file cannot hold -- errors, never wild reads
    p = tmp_path / "evil.pcd"
    head = "VERSION 0.7\nFIELDS x y z w\nSIZE %s\nTYPE %s\nCOUNT 1 1 1 1\nWIDTH 4\nHEIGHT 1\nPOINTS 4\nDATA binary\n"
    p.write_bytes((head % ("4 4 4 -8", "F F F F")).encode() + b"\0" * 64)
    with pytest.raises(api.PclHipError, match="SIZE"):
        api.loadPCDFile(str(p))
    p.write_bytes((head % ("4 4 4 3", "F F F F")).encode() + b"\0" * 64)
    with pytest.raises(api.PclHipError, match="SIZE"):
        api.getPCDHeader(str(p))
    p.write_bytes((head % ("4 4 4 4", "F F F Q")).encode() + b"\0" * 64)
    with pytest.raises(api.PclHipError, match="TYPE"):
        api.getPCDHeader(str(p))
    p.write_bytes((head % ("4 4 4 2", "F F F F")).encode() + b"\0" * 64)
    with pytest.raises(api.PclHipError, match="TYPE F"):
        api.getPCDHeader(str(p))
    p.write_text("VERSION 0.7\nFIELDS x y z\nSIZE 4 4 4\nTYPE F F F\nFIELDS x y z a b c d e f g\nCOUNT 1 1 1 1 1 1 1 1 1 1\n"
                 "WIDTH 1\nHEIGHT 1\nPOINTS 1\nDATA ascii\n0 0 0 0 0 0 0 0 0 0\n")
    with pytest.raises(api.PclHipError, match="COUNT"):
        api.getPCDHeader(str(p))
    p.write_text("VERSION 0.7\nFIELDS x y z\nSIZE 4 4 4\nTYPE F F F\nCOUNT 1 1 -5\nWIDTH 1\nHEIGHT 1\nPOINTS 1\nDATA ascii\n0 0 0\n")
    with pytest.raises(api.PclHipError, match="COUNT"):
        api.getPCDHeader(str(p))
    big = 2 ** 62
    p.write_text("VERSION 0.7\nFIELDS x y z\nSIZE 4 4 4\nTYPE F F F\nCOUNT 1 1 1\nWIDTH %d\nHEIGHT 1\nPOINTS %d\nDATA binary\n" % (1, big))
    with pytest.raises(api.PclHipError):
        api.loadPCDFile(str(p))
    p.write_text("VERSION 0.7\nFIELDS x y z\nSIZE 4 4 4\nTYPE F F F\nCOUNT 1 1 1\nWIDTH 300000000\nHEIGHT 1\nPOINTS 300000000\nDATA ascii\n0 0 0\n")
    with pytest.raises(api.PclHipError, match="shorter"):
        api.loadPCDFile(str(p))
    # binary_compressed: a size word the stream cannot produce (LZF grows by at most 88x) is rejected before anything of
    # that size is allocated -- 4 GB cleared for a 200-byte file took 10-50 s (found by tests/cpp/fuzz_host_io.cpp)
    import struct
    import time
    good = tmp_path / "good.pcd"
    api.savePCDFile(str(good), np.ones((8, 4), np.float32), "binary_compressed")
    raw = good.read_bytes()
    at = raw.index(b"DATA binary_compressed\n") + len(b"DATA binary_compressed\n")
    for usize in (0xFFFFFFFF, 0x7FFFFFFF, 1 << 20):
        p.write_bytes(raw[:at + 4] + struct.pack("<I", usize) + raw[at + 8:])
        t0 = time.perf_counter()
        with pytest.raises(api.PclHipError, match="decompressed lzf"):
            api.loadPCDFile(str(p))
        assert time.perf_counter() - t0 < 0.5


def test_organized_write_viewpoint_and_field_reader(api, tmp_path):
    from oracle import pcd as opcd
    rng = np.random.default_rng(8)
    w, hgt = 64, 48
    cloud = np.zeros((w * hgt, 12), np.float32)
    cloud[:, :3] = rng.normal(0, 2, (w * hgt, 3))
    cloud[:, 3] = 1
    cloud[:, 4:7] = rng.normal(0, 1, (w * hgt, 3))
    cloud[:, 8] = rng.uniform(0, 1, w * hgt)
    vp = [0.5, -1.0, 2.0, 0.5, 0.5, -0.5, 0.5]
    for mode in ("ascii", "binary", "binary_compressed"):
        path = str(tmp_path / ("org_%s.pcd" % mode))
        api.savePCDFile(path, cloud, mode, width=w, height=hgt, viewpoint=vp)
        info = api.getPCDHeader(path)
        assert (info.width, info.height, info.points, info.version) == (w, hgt, w * hgt, 7)
        assert list(info.viewpoint) == vp
        oh, of, _ = opcd.read(path)
        assert (oh["width"], oh["height"]) == (w, hgt) and oh["viewpoint"] == vp
        back, dense = api.loadPCDFile(path, with_normals=True)
        assert dense
        if mode == "ascii":
            assert np.allclose(back, cloud, rtol=1e-7, atol=0)
        else:
            assert np.array_equal(back, cloud)
        for name, col in (("x", 0), ("z", 2), ("normal_y", 5), ("curvature", 8)):
            got = api.loadPCDField(path, name)
            assert np.array_equal(got, back[:, col])
            assert np.array_equal(got, of[name][:, 0].astype(np.float32))
    with pytest.raises(AssertionError):
        api.savePCDFile(str(tmp_path / "bad.pcd"), cloud, "binary", width=10, height=10)
    with pytest.raises(api.PclHipError, match="no field named"):
        api.loadPCDField(path, "intensity")
    # packed colours of a reference file keep their 32 bits; float64 fields convert
    ref = os.path.join(PCD, "colored_cloud.pcd")
    rgb = api.loadPCDField(ref, "rgb")
    oh, of, _ = opcd.read(ref)
    assert np.array_equal(rgb.view(np.uint32), of["rgb"][:, 0].view(np.uint32))
    assert np.array_equal(api.loadPCDField(ref, "curvature"), of["curvature"][:, 0])
    p = tmp_path / "f64.pcd"
    p.write_text("VERSION 0.7\nFIELDS x y z t\nSIZE 4 4 4 8\nTYPE F F F F\nCOUNT 1 1 1 2\nWIDTH 2\nHEIGHT 1\nPOINTS 2\n"
                 "DATA ascii\n0 0 0 1.5 2.5\n1 1 1 3.25 4.75\n")
    assert api.loadPCDField(str(p), "t", 1).tolist() == [2.5, 4.75]
    with pytest.raises(api.PclHipError, match="COUNT"):
        api.loadPCDField(str(p), "t", 2)
