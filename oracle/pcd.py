"""CPU oracle for the PCD reader/writer (test infrastructure only -- see oracle/__init__.py).

Restates io/src/pcd_io.cpp of the reference in numpy / plain Python:
  header  PCDReader::readHeader      :115-392
  ascii   PCDReader::readBodyASCII   :456-559
  binary  PCDReader::readBodyBinary  :561-675 (binary_compressed: u32 compressed size, u32 uncompressed
          size, LZF stream with the fields as struct of arrays)
LZF is liblzf's format (Marc Lehmann; vendored by the reference as io/src/lzf.cpp, which cannot be built
here: pcl_macros.h needs the generated pcl_config.h and Boost).  The decoder below is the published
format: control byte c < 32 -> literal run of c+1 bytes; else back reference, length (c >> 5) + 2 (7 ->
extended by the next byte), distance (((c & 31) << 8) | next byte) + 1.
Pinned on the binary / binary_compressed files the reference ships for its own tests
(tests/golden/pcd/, copied by tests/golden/make_golden.py): the stored uncompressed size (= POINTS x
point size) must be reproduced exactly by the decoder, and the decoded coordinates must be finite and
inside the extents recorded when the fixtures were made.
"""
import struct

import numpy as np

_TYPES = {("F", 4): np.float32, ("F", 8): np.float64, ("I", 1): np.int8, ("I", 2): np.int16,
          ("I", 4): np.int32, ("I", 8): np.int64, ("U", 1): np.uint8, ("U", 2): np.uint16,
          ("U", 4): np.uint32, ("U", 8): np.uint64}


def lzf_decompress(data, out_len):
    out = bytearray()
    i, n = 0, len(data)
    while i < n:
        ctrl = data[i]
        i += 1
        if ctrl < 32:
            out += data[i:i + ctrl + 1]
            i += ctrl + 1
        else:
            length = ctrl >> 5
            if length == 7:
                length += data[i]
                i += 1
            dist = ((ctrl & 0x1F) << 8) + data[i] + 1
            i += 1
            length += 2
            start = len(out) - dist
            if start < 0:
                raise ValueError("LZF back reference before the start of the output")
            for k in range(length):  # may overlap
                out.append(out[start + k])
    if len(out) != out_len:
        raise ValueError("LZF output size %d != %d" % (len(out), out_len))
    return bytes(out)


def read_header(raw):
    """-> dict(fields=[(name, size, type, count)], width, height, points, data, offset, viewpoint)."""
    pos = 0
    h = {"fields": [], "width": 0, "height": 0, "points": 0, "data": "ascii", "offset": 0,
         "viewpoint": [0, 0, 0, 1, 0, 0, 0], "version": 6}
    names, sizes, types, counts = [], None, None, None
    wr = hr = False
    while pos < len(raw):
        eol = raw.find(b"\n", pos)
        if eol < 0:
            eol = len(raw)
        line = raw[pos:eol].decode("latin-1")
        pos = eol + 1
        st = line.split()
        if not st or st[0].startswith("#") or st[0].startswith("VERSION"):
            continue
        k = st[0]
        if k.startswith("FIELDS") or k.startswith("COLUMNS"):
            names = st[1:]
        elif k.startswith("SIZE"):
            sizes = [int(v) for v in st[1:]]
        elif k.startswith("TYPE"):
            types = [v[0] for v in st[1:]]
        elif k.startswith("COUNT"):
            counts = [int(v) for v in st[1:]]
        elif k.startswith("WIDTH"):
            h["width"], wr = int(st[1]), True
        elif k.startswith("HEIGHT"):
            h["height"], hr = int(st[1]), True
        elif k.startswith("VIEWPOINT"):
            h["viewpoint"] = [float(v) for v in st[1:8]]
            h["version"] = 7
        elif k.startswith("POINTS"):
            h["points"] = int(st[1])
        elif k.startswith("DATA"):
            h["data"] = st[1]
            h["offset"] = pos
            break
        else:
            break
    n = len(names)
    sizes = sizes or [4] * n
    types = types or ["F"] * n
    counts = counts or [1] * n
    h["fields"] = [(names[i], sizes[i], types[i], counts[i]) for i in range(n) if counts[i] >= 1]
    if not wr and not hr:
        h["width"], h["height"] = h["points"], 1
    if not hr:
        h["height"] = 1
        if h["width"] == 0:
            h["width"] = h["points"]
    assert h["width"] * h["height"] == h["points"], "HEIGHT x WIDTH != number of points"
    return h


def read(path):
    """-> (header dict, {field name: array [points, count]}, is_dense)"""
    raw = open(path, "rb").read()
    h = read_header(raw)
    n = h["points"]
    fields = h["fields"]
    out = {}
    if h["data"].startswith("ascii"):
        rows = [ln.split() for ln in raw[h["offset"]:].decode("latin-1").split("\n") if ln.strip()][:n]
        col = 0
        for name, size, typ, count in fields:
            dt = _TYPES[(typ, size)]
            arr = np.empty((n, count), dt)
            for i, r in enumerate(rows):
                for c in range(count):
                    tok = r[col + c]
                    arr[i, c] = np.nan if tok in ("nan", "-nan", "NaN") else (float(tok) if typ == "F" else int(tok))
            out[name] = arr
            col += count
    elif h["data"].startswith("binary_compressed"):
        csize, usize = struct.unpack_from("<II", raw, h["offset"])
        soa = lzf_decompress(raw[h["offset"] + 8:h["offset"] + 8 + csize], usize)
        off = 0
        for name, size, typ, count in fields:
            if name == "_":
                continue
            dt = np.dtype(_TYPES[(typ, size)])
            out[name] = np.frombuffer(soa, dt, n * count, off).reshape(n, count).copy()
            off += n * count * size
        assert off == usize, "compressed stream size does not match the fields"
    else:
        step = sum(s * c for _, s, _, c in fields)
        body = np.frombuffer(raw, np.uint8, n * step, h["offset"]).reshape(n, step)
        off = 0
        for name, size, typ, count in fields:
            dt = np.dtype(_TYPES[(typ, size)])
            out[name] = body[:, off:off + size * count].copy().view(dt).reshape(n, count)
            off += size * count
    dense = all(np.isfinite(out[nm]).all() for nm, s, t, c in fields if nm in out and t == "F")
    return h, out, dense


def xyz(path, with_normals=False):
    """PointXYZ-like [n,4] (w = 1) or PointNormal-like [n,12] float32 records, as pclhip_pcd_read fills them."""
    h, f, dense = read(path)
    n = h["points"]
    rec = np.zeros((n, 12 if with_normals else 4), np.float32)
    for k, nm in enumerate("xyz"):
        rec[:, k] = f[nm][:, 0].astype(np.float32)
    rec[:, 3] = 1.0
    if with_normals and all(("normal_" + a) in f for a in "xyz"):
        for k, a in enumerate("xyz"):
            rec[:, 4 + k] = f["normal_" + a][:, 0].astype(np.float32)
        if "curvature" in f:
            rec[:, 8] = f["curvature"][:, 0].astype(np.float32)
    return rec, dense
