"""The reference's streaming genotype backend on disk (`storage=:stream`): 2-bit packed, marker-major `.jgb2`
payload + sidecars, written / read in the reference's own format so that the outputs of the reference's
`prepare_streaming_genotypes` load directly (and vice versa).

Format (src/1.JWAS/src/markers/streaming_genotypes.jl:364-367,600-654,884-971):
    <prefix>.jgb2          nMarkers rows of stride = cld(nObs,4) bytes; individual i (0-based) of a marker sits in
                           byte i>>2 at bit shift (i&3)<<1; codes 0/1/2 = genotype, 3 = missing
    <prefix>.meta          tab-separated key/value manifest (version, *_path, nObs, nMarkers, nMarkersAll,
                           stride_bytes, centered, sum2pq)
    <prefix>.obsid.txt / .markerid.txt     one ID per line
    <prefix>.selected.i32  1-based raw-column index of every kept marker (little-endian Int32)
    <prefix>.mean.f32 / .xpRinvx.f32 / .afreq.f32   per-marker Float32 vectors (little-endian)

`prepare_streaming_genotypes` converts an in-memory matrix, or a delimited text file either by loading it
(`conversion_mode="dense"`, streaming_genotypes.jl:499-656) or in two chunked passes with one row chunk in memory
(`"lowmem"`, the reference's default, :658-817); QC follows get_genotypes (readgenotypes.jl:372-401).
"""
import os

import numpy as np


def pack_2bit(raw_codes):
    """raw_codes: n x p uint8 in {0,1,2,3} -> p x cld(n,4) uint8 payload."""
    n, p = raw_codes.shape
    stride = (n + 3) // 4
    pad = np.zeros((stride * 4, p), dtype=np.uint8)
    pad[:n] = raw_codes
    q = pad.reshape(stride, 4, p)
    payload = q[:, 0] | (q[:, 1] << 2) | (q[:, 2] << 4) | (q[:, 3] << 6)
    return np.ascontiguousarray(payload.T.astype(np.uint8))


def unpack_2bit(payload, n):
    """p x stride payload -> n x p uint8 codes."""
    p, stride = payload.shape
    q = np.empty((stride, 4, p), dtype=np.uint8)
    pt = payload.T
    for k in range(4):
        q[:, k] = (pt >> (2 * k)) & 3
    return q.reshape(stride * 4, p)[:n]


def _check_codes(vals):
    if np.any((vals != np.round(vals)) | (vals < 0) | (vals > 2)):
        raise ValueError("Streaming backend supports only genotype values 0, 1, 2 (and the missing value).")


def _write_sidecars(prefix, n, p_all, selected, mean, afreq, xp, center, obs_ids, marker_all):
    marker_ids = [marker_all[j] for j in selected]
    paths = {k: prefix + ext for k, ext in (("data_path", ".jgb2"), ("obs_path", ".obsid.txt"), ("marker_path", ".markerid.txt"),
                                             ("selected_path", ".selected.i32"), ("mean_path", ".mean.f32"),
                                             ("xp_path", ".xpRinvx.f32"), ("afreq_path", ".afreq.f32"))}
    with open(paths["obs_path"], "w") as fh:
        fh.write("".join(v + "\n" for v in obs_ids))
    with open(paths["marker_path"], "w") as fh:
        fh.write("".join(v + "\n" for v in marker_ids))
    (selected + 1).astype("<i4").tofile(paths["selected_path"])
    mean.astype("<f4").tofile(paths["mean_path"])
    xp.astype("<f4").tofile(paths["xp_path"])
    afreq.astype("<f4").tofile(paths["afreq_path"])
    sum2pq = float((np.float32(2.0) * afreq * (np.float32(1.0) - afreq)).sum(dtype=np.float32))
    entries = [("version", "1")] + list(paths.items()) + [
        ("nObs", str(n)), ("nMarkers", str(len(selected))), ("nMarkersAll", str(p_all)),
        ("stride_bytes", str((n + 3) // 4)), ("centered", "1" if center else "0"), ("sum2pq", repr(sum2pq))]
    with open(prefix + ".meta", "w") as fh:
        fh.write("".join(f"{k}\t{v}\n" for k, v in entries))
    return paths


def _marker_summaries(cnt, s1, s2, n, quality_control, MAF, center, marker_ids=None):
    """per-marker mean / allele frequency / QC selection / x'x from the running sums (readgenotypes.jl:372-401,
    streaming_genotypes.jl:283-285)"""
    if np.any(cnt == 0):                                           # streaming_genotypes.jl:274-278
        j = int(np.flatnonzero(cnt == 0)[0])
        name = marker_ids[j] if marker_ids is not None else str(j + 1)
        raise ValueError(f"Marker {name} has only missing values.")
    mean = (s1 / cnt).astype(np.float32)
    afreq = (mean / np.float32(2.0)).astype(np.float32)
    ss_centered = (s2 - mean.astype(np.float64) * s1).astype(np.float32)
    if quality_control:                                            # strict inequalities in Float32 (:288-290)
        maf32 = np.float32(MAF)
        keep = (maf32 < afreq) & (afreq < np.float32(1.0) - maf32) & (ss_centered != np.float32(0.0))
    else:
        keep = np.ones(cnt.size, dtype=bool)
    selected = np.flatnonzero(keep)
    if selected.size == 0:
        raise ValueError("No markers remain after streaming genotype quality control.")          # :294-296
    mean, afreq = mean[selected], afreq[selected]
    mu = mean.astype(np.float64)
    xp = ((s2[selected] - mu * s1[selected]) if center else (s2[selected] + (n - cnt[selected]) * mu * mu)).astype(np.float32)
    return selected, mean, afreq, xp


def _prepare_from_file_lowmem(path, prefix, *, separator, header, missing_value, quality_control, MAF, center,
                              chunk_rows, disk_guard_ratio):
    """Low-memory conversion of a delimited text file (streaming_genotypes.jl:658-817): two passes over the file in row
    chunks -- (1) per-marker counts / sums for means, QC and x'x; (2) the 2-bit codes of the kept markers written
    straight into the marker-major payload (a memory map, one byte column block per chunk of 4k rows).  Peak memory is
    one chunk (chunk_rows x p floats), not the matrix."""
    import shutil
    import pandas as pd
    with open(path) as fh:
        row1 = [t.strip().strip('"') for t in fh.readline().rstrip("\r\n").split(separator) if t != ""]
    p_all = len(row1) - 1
    marker_all = [str(t) for t in row1[1:]] if header else [str(i + 1) for i in range(p_all)]
    chunk_rows = max(4, (int(chunk_rows) // 4) * 4)                  # whole payload bytes per chunk
    reader = lambda: pd.read_csv(path, sep=separator, header=None, skiprows=1 if header else 0, dtype={0: str}, chunksize=chunk_rows)
    cnt, s1, s2 = np.zeros(p_all, dtype=np.int64), np.zeros(p_all), np.zeros(p_all)
    obs_ids = []
    for chunk in reader():                                           # ---- pass 1
        obs_ids += [str(v) for v in chunk.iloc[:, 0]]
        G = chunk.iloc[:, 1:].to_numpy(dtype=np.float64)
        miss = G == missing_value
        vals = np.where(miss, 0.0, G)
        _check_codes(vals)
        cnt += (~miss).sum(axis=0); s1 += vals.sum(axis=0); s2 += (vals * vals).sum(axis=0)
    n = len(obs_ids)
    selected, mean, afreq, xp = _marker_summaries(cnt, s1, s2, n, quality_control, MAF, center, marker_all)
    stride = (n + 3) // 4
    need = int(len(selected)) * stride
    free = shutil.disk_usage(os.path.dirname(prefix) or ".").free
    if need > disk_guard_ratio * free:                               # streaming_genotypes.jl disk guard
        raise OSError(f"Insufficient disk space for streaming conversion: need {need} bytes, "
                      f"allowed {int(disk_guard_ratio * free)} (disk_guard_ratio={disk_guard_ratio}).")
    mm = np.memmap(prefix + ".jgb2", dtype=np.uint8, mode="w+", shape=(max(len(selected), 1), max(stride, 1)))
    row0 = 0
    for chunk in reader():                                           # ---- pass 2
        G = chunk.iloc[:, 1:].to_numpy(dtype=np.float64)[:, selected]
        codes = np.where(G == missing_value, 3, G).astype(np.uint8)
        block = pack_2bit(codes)                                     # p_sel x cld(rows, 4)
        mm[:len(selected), row0 // 4: row0 // 4 + block.shape[1]] = block
        row0 += codes.shape[0]
    mm.flush()
    del mm
    if len(selected) == 0 or stride == 0:
        open(prefix + ".jgb2", "wb").close()
    _write_sidecars(prefix, n, p_all, selected, mean, afreq, xp, center, obs_ids, marker_all)
    return prefix


def prepare_streaming_genotypes(genotypes, output_prefix=None, *, obs_ids=None, marker_ids=None, missing_value=9.0,
                                quality_control=True, MAF=0.01, center=True, separator=",", header=True,
                                conversion_mode="lowmem", auto_dense_max_bytes=2 ** 30, chunk_rows=4096,
                                disk_guard_ratio=0.9):
    """genotypes: n x p array of 0/1/2 (missing_value = missing), or the path of a delimited text file (first column =
    individual IDs).  Writes <output_prefix>.{jgb2,meta,...} and returns the prefix (streaming_genotypes.jl:819-877).
    File input: conversion_mode = "lowmem" (default, like the reference; two chunked passes, memory = one row chunk),
    "dense" (load the matrix) or "auto" (dense when n*p*4 <= auto_dense_max_bytes)."""
    if isinstance(genotypes, (str, os.PathLike)):
        import pandas as pd
        path = str(genotypes)
        if conversion_mode not in ("lowmem", "dense", "auto"):
            raise ValueError("conversion_mode must be :auto, :dense, or :lowmem.")
        if auto_dense_max_bytes < 0:
            raise ValueError("auto_dense_max_bytes must be non-negative.")
        prefix = os.path.abspath(str(output_prefix if output_prefix is not None else os.path.splitext(path)[0] + "_stream"))
        for ext in (".meta", ".jgb2"):
            if prefix.endswith(ext):
                prefix = prefix[:-len(ext)]
        os.makedirs(os.path.dirname(prefix) or ".", exist_ok=True)
        mode = conversion_mode
        if mode == "auto":
            with open(path) as fh:
                ncol = len([t for t in fh.readline().rstrip("\r\n").split(separator) if t != ""]) - 1
                nrow = sum(1 for _ in fh) + (0 if header else 1)
            est = nrow * ncol * 4
            mode = "dense" if est <= auto_dense_max_bytes else "lowmem"
            print(f"Auto conversion mode selected :{mode} (estimated dense bytes={est}, auto_dense_max_bytes={auto_dense_max_bytes}).")
        if mode == "lowmem":
            return _prepare_from_file_lowmem(path, prefix, separator=separator, header=header, missing_value=missing_value,
                                             quality_control=quality_control, MAF=MAF, center=center, chunk_rows=chunk_rows,
                                             disk_guard_ratio=disk_guard_ratio)
        with open(path) as fh:
            row1 = [t.strip().strip('"') for t in fh.readline().rstrip("\r\n").split(separator) if t != ""]
        tab = pd.read_csv(path, sep=separator, header=None, skiprows=1 if header else 0, dtype={0: str})
        return prepare_streaming_genotypes(tab.iloc[:, 1:].to_numpy(dtype=np.float64), prefix,
                                           obs_ids=[str(v) for v in tab.iloc[:, 0]],
                                           marker_ids=[str(t) for t in row1[1:]] if header else None,
                                           missing_value=missing_value, quality_control=quality_control, MAF=MAF, center=center)
    if output_prefix is None:
        raise ValueError("output_prefix is required for in-memory genotypes")
    G = np.asarray(genotypes)
    n, p_all = G.shape
    miss = (G == missing_value)
    vals = np.where(miss, 0, G).astype(np.float64)
    if np.any((vals != np.round(vals)) | (vals < 0) | (vals > 2)):
        raise ValueError("Streaming backend supports only genotype values 0, 1, 2 (and the missing value).")
    cnt = (~miss).sum(axis=0)
    s1 = vals.sum(axis=0)
    s2 = (vals * vals).sum(axis=0)
    # means, QC selection and x'x exactly as the file path computes them (x'x of the centred, mean-imputed column:
    # sum v^2 - mu*sum v over non-missing, streaming_genotypes.jl:283-285)
    selected, mean, afreq, xp = _marker_summaries(cnt, s1, s2, n, quality_control, MAF, center, marker_ids)
    codes = np.where(miss[:, selected], 3, vals[:, selected]).astype(np.uint8)
    payload = pack_2bit(codes)
    prefix = os.path.abspath(str(output_prefix))
    for ext in (".meta", ".jgb2"):
        if prefix.endswith(ext):
            prefix = prefix[:-len(ext)]
    os.makedirs(os.path.dirname(prefix) or ".", exist_ok=True)
    obs_ids = [str(i + 1) for i in range(n)] if obs_ids is None else [str(v) for v in obs_ids]
    marker_all = [f"m{j + 1}" for j in range(p_all)] if marker_ids is None else [str(v) for v in marker_ids]
    marker_ids = [marker_all[j] for j in selected]
    paths = {k: prefix + ext for k, ext in (("data_path", ".jgb2"), ("obs_path", ".obsid.txt"), ("marker_path", ".markerid.txt"),
                                             ("selected_path", ".selected.i32"), ("mean_path", ".mean.f32"),
                                             ("xp_path", ".xpRinvx.f32"), ("afreq_path", ".afreq.f32"))}
    payload.tofile(paths["data_path"])
    with open(paths["obs_path"], "w") as fh:
        fh.write("".join(v + "\n" for v in obs_ids))
    with open(paths["marker_path"], "w") as fh:
        fh.write("".join(v + "\n" for v in marker_ids))
    (selected + 1).astype("<i4").tofile(paths["selected_path"])
    mean.astype("<f4").tofile(paths["mean_path"])
    xp.astype("<f4").tofile(paths["xp_path"])
    afreq.astype("<f4").tofile(paths["afreq_path"])
    sum2pq = float((np.float32(2.0) * afreq * (np.float32(1.0) - afreq)).sum(dtype=np.float32))
    entries = [("version", "1")] + list(paths.items()) + [
        ("nObs", str(n)), ("nMarkers", str(len(selected))), ("nMarkersAll", str(p_all)),
        ("stride_bytes", str((n + 3) // 4)), ("centered", "1" if center else "0"), ("sum2pq", repr(sum2pq))]
    with open(prefix + ".meta", "w") as fh:
        fh.write("".join(f"{k}\t{v}\n" for k, v in entries))
    return prefix


def resolve_prefix(path):
    path = os.path.abspath(str(path))
    for ext in (".meta", ".jgb2"):                                  # _resolve_streaming_prefix (:97-105)
        if path.endswith(ext):
            return path[:-len(ext)]
    return path


def load_streaming_backend(path):
    """Host-side metadata of a streaming backend (load_streaming_backend, streaming_genotypes.jl:884-971): everything
    except the payload, which goes straight from the file to HBM (jwas_hip_load_jgb2)."""
    prefix = resolve_prefix(path)
    meta_path = prefix + ".meta"
    if not os.path.isfile(meta_path):
        raise FileNotFoundError(f"Streaming manifest is not found: {meta_path}")
    meta = {}
    with open(meta_path) as fh:
        for line in fh:
            parts = line.rstrip("\n").split("\t", 1)
            if len(parts) == 2:
                meta[parts[0]] = parts[1]
    n, p, stride = int(meta["nObs"]), int(meta["nMarkers"]), int(meta["stride_bytes"])

    def side(key, ext):
        f = meta.get(key, "")
        return f if f and os.path.isfile(f) else prefix + ext

    data_path = side("data_path", ".jgb2")
    if os.path.getsize(data_path) != p * stride:
        raise ValueError(f"Packed genotype file size does not match metadata for {data_path}")
    obs = [v for v in open(side("obs_path", ".obsid.txt")).read().split("\n") if v]
    markers = [v for v in open(side("marker_path", ".markerid.txt")).read().split("\n") if v]
    if len(obs) != n:
        raise ValueError("Number of IDs does not match nObs in manifest.")
    if len(markers) != p:
        raise ValueError("Number of markers does not match nMarkers in manifest.")
    return {
        "prefix": prefix, "data_path": data_path, "nObs": n, "nMarkers": p, "stride_bytes": stride,
        "nMarkersAll": int(meta.get("nMarkersAll", p)), "centered": int(meta["centered"]) == 1,
        "sum2pq": float(meta["sum2pq"]), "obsID": obs, "markerID": markers,
        "marker_means": np.fromfile(side("mean_path", ".mean.f32"), dtype="<f4", count=p),
        "xpRinvx": np.fromfile(side("xp_path", ".xpRinvx.f32"), dtype="<f4", count=p),
        "allele_freq": np.fromfile(side("afreq_path", ".afreq.f32"), dtype="<f4", count=p),
    }


def decode_markers(backend, j0=0, count=None):
    """CPU decode of markers [j0, j0+count) (decode_marker!, :978-1002) -- small inputs / checks only."""
    p, stride, n = backend["nMarkers"], backend["stride_bytes"], backend["nObs"]
    count = p - j0 if count is None else count
    payload = np.fromfile(backend["data_path"], dtype=np.uint8, count=count * stride, offset=j0 * stride).reshape(count, stride)
    codes = unpack_2bit(payload, n)
    mu = backend["marker_means"][j0:j0 + count].astype(np.float32)
    v = np.where(codes == 3, mu[None, :], codes.astype(np.float32)).astype(np.float32)
    return np.asfortranarray(v - mu[None, :] if backend["centered"] else v)
