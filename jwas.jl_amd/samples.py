"""Binary marker-effect sample files.

The reference writes every saved marker-effect sample as one comma-separated text row of p values
(MCMC_samples_marker_effects_<geno>_<trait>.txt, output.jl:443-526): 600 000 columns, ~6 MB of text per saved sample at
config 2, almost all of them "0.0".  Here a saved sample is a sparse record -- the nonzero effects compacted on the device
(jwas_hip_get_alpha_sparse) -- appended to `<same name>.bin`:

    header   8 bytes  magic b"JWASMES1"
             int64    p (number of markers)
             int64    length L of the marker-ID block, then L bytes: the IDs joined by "\\n" (UTF-8)
    record   int32    nnz
             int32    idx[nnz]   0-based marker indices, ascending
             float32  val[nnz]

`to_text` converts a file to the reference's text layout (header row of marker IDs, one row per sample), so every
consumer of the reference's files keeps working; `read_dense` / `iter_records` feed the window GWAS directly.
"""
import struct

import numpy as np

MAGIC = b"JWASMES1"


class MarkerSampleWriter:
    def __init__(self, path, marker_ids):
        self.path = path
        self.p = len(marker_ids)
        self.nsamples = 0
        ids = "\n".join(str(m) for m in marker_ids).encode("utf-8")
        self._fh = open(path, "wb")
        self._fh.write(MAGIC + struct.pack("<qq", self.p, len(ids)) + ids)

    def append(self, idx, val):
        idx = np.ascontiguousarray(idx, dtype="<i4")
        val = np.ascontiguousarray(val, dtype="<f4")
        if idx.shape != val.shape or idx.ndim != 1:
            raise ValueError("idx and val must be 1-D arrays of the same length")
        if idx.size and (idx.min() < 0 or idx.max() >= self.p or np.any(np.diff(idx) <= 0)):
            raise ValueError("marker indices must be ascending and inside [0, p)")
        self._fh.write(struct.pack("<i", idx.size))
        self._fh.write(idx.tobytes())
        self._fh.write(val.tobytes())
        self.nsamples += 1

    def close(self):
        if self._fh is not None:
            self._fh.close()
            self._fh = None


def _open(path):
    fh = open(path, "rb")
    if fh.read(8) != MAGIC:
        fh.close()
        raise ValueError(f"{path} is not a binary marker-effect sample file")
    p, L = struct.unpack("<qq", fh.read(16))
    ids = fh.read(L).decode("utf-8").split("\n") if L else []
    return fh, p, ids


def iter_records(path):
    """Yields (idx int32, val float32) per saved sample."""
    fh, p, _ = _open(path)
    with fh:
        while True:
            head = fh.read(4)
            if len(head) < 4:
                return
            nnz = struct.unpack("<i", head)[0]
            idx = np.frombuffer(fh.read(4 * nnz), dtype="<i4")
            val = np.frombuffer(fh.read(4 * nnz), dtype="<f4")
            if idx.size != nnz or val.size != nnz:
                raise ValueError(f"{path} is truncated")
            yield idx, val


def marker_ids(path):
    fh, p, ids = _open(path)
    fh.close()
    return ids


def read_dense(path):
    """(samples x p float64 matrix, marker IDs) -- what reading the reference's text file gives."""
    fh, p, ids = _open(path)
    fh.close()
    rows = []
    for idx, val in iter_records(path):
        row = np.zeros(p)
        row[idx] = val
        rows.append(row)
    return (np.stack(rows) if rows else np.zeros((0, p))), ids


def to_text(path, text_path):
    """Write the reference's layout (output.jl:320-333,467): header row of marker IDs, one comma-separated row per sample."""
    fh, p, ids = _open(path)
    fh.close()
    with open(text_path, "w") as out:
        out.write(",".join(ids) + "\n")
        for idx, val in iter_records(path):
            row = np.zeros(p, dtype=np.float32)
            row[idx] = val
            row.tofile(out, sep=",", format="%.9g")
            out.write("\n")
    return text_path
