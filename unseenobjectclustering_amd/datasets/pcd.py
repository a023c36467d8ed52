"""PCD (Point Cloud Data) reader — what the reference gets from `pcl.load(f).to_array()` in its OCID / OSD loaders
(/root/reference/lib/datasets/ocid_object.py:105, osd_object.py:92): the x, y, z fields of every point as a
float32 [POINTS, 3] array, NaNs left in place (the loaders zero them).  python-pcl is not available in this build;
the three storage modes of the format (ascii, binary, binary_compressed) are parsed here, the LZF stream of the
compressed mode by the native library (uoc_lzf_decompress)."""
from __future__ import annotations

import ctypes

import numpy as np

_NP = {("F", 4): np.float32, ("F", 8): np.float64, ("U", 1): np.uint8, ("U", 2): np.uint16, ("U", 4): np.uint32,
       ("I", 1): np.int8, ("I", 2): np.int16, ("I", 4): np.int32}


def _header(f):
    h = {}
    while True:
        line = f.readline()
        if not line:
            raise ValueError("PCD: no DATA line")
        line = line.decode("ascii", "replace").strip()
        if not line or line.startswith("#"):
            continue
        key, _, rest = line.partition(" ")
        h[key.upper()] = rest.split()
        if key.upper() == "DATA":
            return h


def load_xyz(path: str) -> np.ndarray:
    """[POINTS, 3] float32 (x, y, z) of a .pcd file; other fields (rgb, normals...) are skipped."""
    with open(path, "rb") as f:
        h = _header(f)
        fields = [s.lower() for s in h["FIELDS"]]
        sizes = [int(s) for s in h["SIZE"]]
        types = h["TYPE"]
        counts = [int(c) for c in h.get("COUNT", ["1"] * len(fields))]
        npts = int(h["POINTS"][0]) if "POINTS" in h else int(h["WIDTH"][0]) * int(h["HEIGHT"][0])
        mode = h["DATA"][0].lower()
        for k in ("x", "y", "z"):
            if k not in fields:
                raise ValueError(f"PCD {path}: no '{k}' field")
        dtype = np.dtype([(fields[i] if counts[i] == 1 else fields[i], _NP[(types[i], sizes[i])], (counts[i],) if counts[i] > 1 else ())
                          for i in range(len(fields))])
        if mode == "ascii":
            table = np.loadtxt(f, dtype=np.float64, ndmin=2)
            if table.shape[0] != npts:
                raise ValueError(f"PCD {path}: {table.shape[0]} rows, header says {npts}")
            col = np.cumsum([0] + counts)
            return np.stack([table[:, col[fields.index(k)]] for k in ("x", "y", "z")], axis=1).astype(np.float32)
        if mode == "binary":
            rec = np.frombuffer(f.read(npts * dtype.itemsize), dtype=dtype, count=npts)
            return np.stack([rec[k].astype(np.float32) for k in ("x", "y", "z")], axis=1)
        if mode == "binary_compressed":
            from .. import _native
            head = np.frombuffer(f.read(8), dtype=np.uint32)
            comp, raw = int(head[0]), int(head[1])
            src = f.read(comp)
            if len(src) != comp or raw != npts * dtype.itemsize:
                raise ValueError(f"PCD {path}: compressed block sizes do not match the header")
            dst = np.empty(raw, dtype=np.uint8)
            n = _native.lib().uoc_lzf_decompress(ctypes.c_char_p(src), comp, ctypes.c_void_p(dst.ctypes.data), raw)
            if n != raw:
                raise ValueError(f"PCD {path}: LZF stream decoded to {n} bytes, expected {raw}: "
                                 + _native.lib().uoc_last_error().decode("utf-8", "replace"))
            out, off = {}, 0       # structure of arrays: all x, then all y, ...
            for i, name in enumerate(fields):
                nbytes = sizes[i] * counts[i] * npts
                if name in ("x", "y", "z"):
                    out[name] = dst[off:off + nbytes].view(_NP[(types[i], sizes[i])]).astype(np.float32)
                off += nbytes
            return np.stack([out["x"], out["y"], out["z"]], axis=1)
        raise ValueError(f"PCD {path}: unknown DATA mode {mode!r}")
