"""GWAS post-processing of the marker-effect samples (reference: src/3.GWAS/src/GWAS.jl).

`GWAS(marker_effects_file)`: model frequency of every marker (GWAS.jl:6-18) -- file I/O only.

`GWAS(model, map_file, marker_effects_file..., window_size=..., ...)`: window posterior probability of association
(GWAS.jl:49-197).  Its cost in the reference is one dense X*alpha plus one product per window for every saved sample,
O(samples x n x p) over the same genotype matrix the sweep streams; here every sample is one launch over its nonzero effects
only (`jwas_hip_window_sums`: sum and sum of squares of each window's genomic values, fp64), the windows and the summary
statistics are assembled on the host exactly as the reference does; the window genetic covariance / correlation of two
traits (`genetic_correlation=true`, GWAS.jl:199-237) uses `jwas_hip_window_sums2`.  Local EBVs stay on the reference."""
import os

import numpy as np


def _read_samples(path, header=True):
    import pandas as pd
    if str(path).endswith(".bin"):                  # binary sparse sample records (samples.py)
        from .samples import read_dense
        return read_dense(path)
    tab = pd.read_csv(path, header=0 if header else None)
    ids = [str(c) for c in tab.columns] if header else list(range(1, tab.shape[1] + 1))
    return tab.to_numpy(dtype=np.float64), ids


def model_frequency(marker_effects_file, header=True):
    """GWAS.jl:6-18"""
    import pandas as pd
    print("Compute the model frequency for each marker (the probability the marker is included in the model).")
    samples, ids = _read_samples(marker_effects_file, header)
    return pd.DataFrame({"marker_ID": ids, "modelfrequency": (samples != 0.0).mean(axis=0)})


def build_windows(chr_, pos, window_size_bp, sliding_window):
    """GWAS.jl:90-134: windows as [column_start, column_end) over the (sorted) map; empty windows are dropped."""
    out = dict(chr=[], pos_start=[], pos_end=[], snp_start=[], snp_end=[], col_start=[], col_end=[], nsnp=[])
    index_start = 0
    chr_ = np.asarray(chr_)
    pos = np.asarray(pos, dtype=np.int64)
    seen = []
    for c in chr_:
        if c not in seen:
            seen.append(c)
    for c in seen:
        pc = pos[chr_ == c]
        if not sliding_window:
            nwin = int(np.ceil(pc[-1] / window_size_bp))
        else:
            nwin = int(np.argmax(pc >= pc[-1] - window_size_bp)) + 1          # findfirst(x -> x >= last - size)
        for j in range(nwin):
            start = window_size_bp * j if not sliding_window else int(pc[j])
            end = start + window_size_bp
            inwin = (start <= pc) & (pc < end)
            k = int(inwin.sum())
            if k:
                first, last = int(np.argmax(inwin)), int(len(pc) - 1 - np.argmax(inwin[::-1]))
                out["snp_start"].append(int(pc[first])); out["snp_end"].append(int(pc[last]))
                out["col_start"].append(index_start); out["col_end"].append(index_start + k)
                out["chr"].append(str(c)); out["pos_start"].append(start); out["pos_end"].append(end); out["nsnp"].append(k)
            index_start += k if not sliding_window else 1
    return out


def GWAS(*args, window_size="1 Mb", sliding_window=False, GWAS=True, threshold=0.001, genetic_correlation=False,
         local_EBV=False, header=True, output_winVarProps=False, output_folder=".", device=0, _engine=None):
    """GWAS(marker_effects_file; header) or GWAS(model | genotype matrix, map_file, marker_effects_file...; ...)."""
    import pandas as pd
    if len(args) == 1:
        return model_frequency(args[0], header)
    if len(args) < 3:
        raise TypeError("GWAS(model, map_file, marker_effects_file...)")
    mme, map_file, files = args[0], args[1], list(args[2:])
    if local_EBV:
        raise NotImplementedError("local_EBV stays on the reference")
    if genetic_correlation and len(files) != 2:
        raise ValueError("genetic_correlation=true needs exactly two marker_effects_files (one per trait).")
    if isinstance(window_size, str):
        parts = window_size.split()
        if len(parts) != 2 or parts[1] != "Mb":
            raise ValueError('The format for window_size is "1 Mb".')
    # ---- the genotypes of the individuals the reference uses here: Mi.output_genotypes (GWAS.jl:148)
    if isinstance(mme, np.ndarray):
        X, marker_ids = np.asarray(mme, dtype=np.float32), None
    else:
        Mi = mme.M[0]
        rows = getattr(Mi, "output_rows", None)
        X = Mi.genotypes if rows is None else Mi.genotypes[rows, :]
        marker_ids = list(Mi.markerID)
    nmarkers = X.shape[1]
    if map_file is False and isinstance(window_size, (int, np.integer)):
        print(f"The map file is not provided. A fake map file is generated with {window_size} markers in each 1 Mb window.")
        step = 1_000_000 / window_size
        mapfile = pd.DataFrame({"markerID": [str(i) for i in range(1, nmarkers + 1)], "chromosome": "1",
                                "position": np.floor(1 + step * np.arange(nmarkers)).astype(np.int64)})
        window_size = "1 Mb"
    else:
        mapfile = pd.read_csv(map_file, header=0 if header else None, dtype={0: str, 1: str})
        mapfile.columns = ["markerID", "chromosome", "position"][:3] + list(mapfile.columns[3:])
        if marker_ids is not None:                              # drop SNPs not used in the analysis (GWAS.jl:79-85)
            keep = mapfile["markerID"].isin(set(marker_ids))
            mapfile = mapfile[keep].reset_index(drop=True)
            if len(mapfile) == 0:
                raise ValueError("Please check the 1st column of the mapfile (i.e., marker ID)")
    window_size_bp = int(float(window_size.split()[0]) * 1_000_000)
    win = build_windows(mapfile["chromosome"].to_numpy(), mapfile["position"].to_numpy(dtype=np.int64), window_size_bp, sliding_window)
    nwin = len(win["nsnp"])
    if not GWAS and not genetic_correlation:
        return tuple()
    if GWAS:
        print(f"Compute the posterior probability of association of the genomic window that explains more than {threshold} "
              "of the total genetic variance.")
    engine = _engine                              # (private: the test-suite injects a sweep engine)
    own = engine is None
    if own:
        from .engine import HipEngine
        engine = HipEngine(device)
    engine.load_dense(np.asfortranarray(X, dtype=np.float32))
    n = X.shape[0]
    cs, ce = np.asarray(win["col_start"]), np.asarray(win["col_end"])
    out, props_out = [], []
    try:
        for fi, path in enumerate(files if GWAS else [], start=1):
            samples, _ = _read_samples(path, True)
            nsamples = samples.shape[0]
            winVar = np.zeros((nsamples, nwin))
            winVarProps = np.zeros((nsamples, nwin))
            for i in range(nsamples):
                a = samples[i].astype(np.float32)
                nz = np.flatnonzero(a)
                # window 0 = all markers (genVar), then every window's own nonzero effects
                lo, hi = np.searchsorted(nz, cs), np.searchsorted(nz, ce)
                counts = hi - lo
                wptr = np.concatenate([[0, nz.size], nz.size + np.cumsum(counts)]).astype(np.int32)
                gather = np.concatenate([nz] + [nz[l:h] for l, h in zip(lo, hi) if h > l]) if nz.size else nz
                s, q = engine.window_sums(wptr, gather, a[gather])
                var = (q - s * s / n) / (n - 1)                  # var(BV) = sample variance (GWAS.jl:153,158)
                winVar[i] = var[1:]
                with np.errstate(divide="ignore", invalid="ignore"):
                    winVarProps[i] = var[1:] / var[0]
            np.savetxt(os.path.join(output_folder, f"MCMC_samples_local_genomic_variance{fi}.txt"), winVar, delimiter=",")
            winVarProps[np.isnan(winVarProps)] = 0.0             # no marker in the model in that sample
            WPPA = (winVarProps > threshold).mean(axis=0)
            prop = np.round(winVarProps.mean(axis=0) * 100, 6)
            order = np.argsort(-WPPA, kind="stable")
            tab = pd.DataFrame({
                "trait": fi, "window": (np.arange(nwin) + 1)[order], "chr": np.asarray(win["chr"])[order],
                "wStart": np.asarray(win["pos_start"])[order], "wEnd": np.asarray(win["pos_end"])[order],
                "start_SNP": np.asarray(win["snp_start"])[order], "end_SNP": np.asarray(win["snp_end"])[order],
                "numSNP": np.asarray(win["nsnp"])[order], "estimateGenVar": winVar.mean(axis=0)[order],
                "stdGenVar": (winVar.std(axis=0, ddof=1) if nsamples > 1 else np.full(nwin, np.nan))[order],
                "prGenVar": prop[order], "WPPA": WPPA[order],
                "PPA_t": np.cumsum(WPPA[order]) / np.arange(1, nwin + 1)})
            tab.to_csv(os.path.join(output_folder, "GWAS_" + str(path).replace("/", "_")), index=False)
            out.append(tab)
            if output_winVarProps:
                props_out.append(winVarProps)
        if genetic_correlation:                                  # GWAS.jl:199-237
            s1, _ = _read_samples(files[0], True)
            s2, _ = _read_samples(files[1], True)
            nsamples = s1.shape[0]
            gcov, gcor = np.zeros((nsamples, nwin)), np.zeros((nsamples, nwin))
            for i in range(nsamples):
                a1, a2 = s1[i].astype(np.float32), s2[i].astype(np.float32)
                nz = np.flatnonzero((a1 != 0) | (a2 != 0))
                lo, hi = np.searchsorted(nz, cs), np.searchsorted(nz, ce)
                wptr = np.concatenate([[0], np.cumsum(hi - lo)]).astype(np.int32)
                gather = np.concatenate([nz[l:h] for l, h in zip(lo, hi) if h > l]) if wptr[-1] else nz[:0]
                su1, q1, su2, q2, cr = engine.window_sums2(wptr, gather, a1[gather], a2[gather])
                cov = (cr - su1 * su2 / n) / (n - 1)
                v1, v2 = (q1 - su1 * su1 / n) / (n - 1), (q2 - su2 * su2 / n) / (n - 1)
                gcov[i] = cov
                with np.errstate(divide="ignore", invalid="ignore"):
                    gcor[i] = cov / np.sqrt(v1 * v2)
            gcov[np.isnan(gcov)] = 0.0
            gcor[~np.isfinite(gcor)] = 0.0
            np.savetxt(os.path.join(output_folder, "MCMC_samples_local_genomic_covariance.txt"), gcov, delimiter=",")
            sd = (lambda m: m.std(axis=0, ddof=1) if nsamples > 1 else np.full(nwin, np.nan))
            tab = pd.DataFrame({
                "trait": "cor(t1,t2)", "window": np.arange(nwin) + 1, "chr": win["chr"], "wStart": win["pos_start"],
                "wEnd": win["pos_end"], "start_SNP": win["snp_start"], "end_SNP": win["snp_end"], "numSNP": win["nsnp"],
                "estimate_cov": gcov.mean(axis=0), "std_cov": sd(gcov), "estimate_cor": gcor.mean(axis=0), "std_cor": sd(gcor)})
            tab.to_csv(os.path.join(output_folder, "GWAS_" + str(files[0]).replace("/", "_") + "_" + str(files[1]).replace("/", "_")), index=False)
            out.append(tab)
    finally:
        if own:
            engine.close()
    return (tuple(out), tuple(props_out)) if output_winVarProps else tuple(out)
