"""Window-based GWAS (src/3.GWAS/src/GWAS.jl) over marker-effect samples: host logic against a literal restatement of
the reference's loops (CPU, numpy stand-in for the device call), the device kernel against numpy, and the reference's own
test sets (test/unit/test_gwas_windows.jl, test/runtests.jl:327-349) on its demo_7animals data."""
import os

import numpy as np
import pandas as pd
import pytest

from jwas_jl_amd import api
from jwas_jl_amd.gwas import GWAS, build_windows
from oracle_engine import OracleEngine

DEMO = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "demo_7animals")


def _literal_gwas(X, samples, col_start, col_end, threshold):
    """GWAS.jl:152-176, as written there"""
    ns, nw = samples.shape[0], len(col_start)
    winVar, props = np.zeros((ns, nw)), np.zeros((ns, nw))
    for i in range(ns):
        a = samples[i]
        genVar = np.var(X @ a, ddof=1)
        for w in range(nw):
            v = np.var(X[:, col_start[w]:col_end[w]] @ a[col_start[w]:col_end[w]], ddof=1)
            winVar[i, w] = v
            props[i, w] = v / genVar if genVar != 0 else np.nan
    props[np.isnan(props)] = 0.0
    return winVar, props, (props > threshold).mean(axis=0)


def _case(tmp_path, seed=0, n=60, p=40, ns=25):
    rng = np.random.default_rng(seed)
    X = rng.integers(0, 3, size=(n, p)).astype(np.float32)
    X -= X.mean(0)
    samples = np.where(rng.random((ns, p)) < 0.15, rng.standard_normal((ns, p)), 0.0).astype(np.float32)
    samples[3] = 0.0                                     # a sample with no marker in the model
    ids = [f"m{j + 1}" for j in range(p)]
    f = tmp_path / "MCMC_samples_marker_effects_geno_y1.txt"
    pd.DataFrame(samples, columns=ids).to_csv(f, index=False, float_format="%.9g")
    chrom = np.repeat(["1", "2", "3"], [15, 15, 10])
    pos = np.concatenate([np.sort(rng.integers(1, 4_000_000, 15)), np.sort(rng.integers(1, 3_000_000, 15)), np.sort(rng.integers(1, 2_500_000, 10))])
    mapf = tmp_path / "map.txt"
    pd.DataFrame({"markerID": ids, "chromosome": chrom, "position": pos}).to_csv(mapf, index=False)
    return X, samples, str(f), str(mapf), chrom, pos


def test_model_frequency_form(tmp_path):
    X, samples, f, mapf, _, _ = _case(tmp_path)
    tab = GWAS(f)
    assert list(tab.columns) == ["marker_ID", "modelfrequency"] and len(tab) == samples.shape[1]
    np.testing.assert_allclose(tab["modelfrequency"], (samples != 0).mean(0))


@pytest.mark.parametrize("sliding", [False, True])
def test_window_gwas_matches_literal_restatement(tmp_path, sliding):
    X, samples, f, mapf, chrom, pos = _case(tmp_path, seed=3)
    res, props = GWAS(X, mapf, f, window_size="1 Mb", sliding_window=sliding, threshold=0.05, output_winVarProps=True,
                      output_folder=str(tmp_path), _engine=OracleEngine("dense"))
    tab = res[0]
    assert list(tab.columns) == ["trait", "window", "chr", "wStart", "wEnd", "start_SNP", "end_SNP", "numSNP", "estimateGenVar",
                                 "stdGenVar", "prGenVar", "WPPA", "PPA_t"]
    win = build_windows(chrom, pos, 1_000_000, sliding)
    winVar, lit_props, wppa = _literal_gwas(X.astype(np.float64), samples.astype(np.float64), win["col_start"], win["col_end"], 0.05)
    np.testing.assert_allclose(props[0], lit_props, atol=1e-9)
    by_window = tab.sort_values("window")
    np.testing.assert_allclose(by_window["WPPA"], wppa)
    np.testing.assert_allclose(by_window["estimateGenVar"], winVar.mean(0), rtol=1e-9)
    np.testing.assert_allclose(by_window["numSNP"], win["nsnp"])
    assert (np.diff(tab["WPPA"]) <= 1e-15).all()                     # sorted by WPPA, descending
    np.testing.assert_allclose(tab["PPA_t"], np.cumsum(tab["WPPA"]) / np.arange(1, len(tab) + 1))
    assert os.path.exists(tmp_path / "MCMC_samples_local_genomic_variance1.txt")
    if not sliding:                                                   # non-overlapping windows partition the markers
        assert by_window["numSNP"].sum() == X.shape[1]


def test_window_size_format_and_fake_map(tmp_path):
    X, samples, f, mapf, _, _ = _case(tmp_path)
    with pytest.raises(ValueError, match='"1 Mb"'):
        GWAS(X, mapf, f, window_size="1 kb", _engine=OracleEngine("dense"), output_folder=str(tmp_path))
    res = GWAS(X, False, f, window_size=8, _engine=OracleEngine("dense"), output_folder=str(tmp_path))   # 8 markers per window
    assert list(res[0].sort_values("window")["numSNP"]) == [8, 8, 8, 8, 8]


@pytest.mark.gpu
def test_device_window_sums_match_numpy():
    import jwas_jl_amd as J
    rng = np.random.default_rng(1)
    n, p = 700, 300
    X = rng.standard_normal((n, p)).astype(np.float32)
    a = np.where(rng.random(p) < 0.2, rng.standard_normal(p), 0).astype(np.float32)
    nz = np.flatnonzero(a)
    wptr = np.array([0, nz.size, nz.size + 5, nz.size + 5, nz.size + 12], dtype=np.int32)      # all | 5 | empty | 7
    idx = np.concatenate([nz, nz[:5], nz[3:10]]).astype(np.int32)
    e = J.HipEngine(0); e.load_dense(X)
    s, q = e.window_sums(wptr, idx, a[idx])
    o = OracleEngine("dense"); o.load_dense(X)
    so, qo = o.window_sums(wptr, idx, a[idx])
    np.testing.assert_allclose(s, so, rtol=1e-12, atol=1e-9)
    np.testing.assert_allclose(q, qo, rtol=1e-12)
    assert s[2] == 0.0 and q[2] == 0.0
    e.load_output_dense(X[:130])
    s2, q2 = e.window_sums(wptr, idx, a[idx], use_output_rows=True)
    o.load_output_dense(X[:130])
    so2, qo2 = o.window_sums(wptr, idx, a[idx], use_output_rows=True)
    np.testing.assert_allclose(q2, qo2, rtol=1e-12)
    with pytest.raises(J.JwasHipError, match="out of range"):
        e.window_sums(np.array([0, 1], dtype=np.int32), np.array([p], dtype=np.int32), np.array([1.0], dtype=np.float32))
    e.close()


@pytest.mark.gpu
def test_reference_gwas_test_sets_on_demo_data(tmp_path):
    """test/unit/test_gwas_windows.jl:9-57 and test/runtests.jl:333-349"""
    pheno = pd.read_csv(os.path.join(DEMO, "phenotypes.txt"), na_values=["NA"], dtype={"ID": str})
    geno = api.get_genotypes(os.path.join(DEMO, "genotypes.txt"), 1.0, separator=",", method="BayesC")
    model = api.build_model("y1 = intercept + geno", 1.0)
    api.outputEBV(model, geno.obsID)
    folder = tmp_path / "test_gwas_win"
    api.runMCMC(model, pheno, chain_length=100, burnin=20, output_samples_frequency=10, outputEBV=True, output_folder=str(folder), seed=123)
    marker_file = str(folder / "MCMC_samples_marker_effects_geno_y1.txt")
    mapfile = os.path.join(DEMO, "map.txt")
    mf = api.GWAS(marker_file)
    assert len(mf) > 0 and {"marker_ID", "modelfrequency"} <= set(mf.columns) and mf["modelfrequency"].between(0, 1).all()
    res = api.GWAS(model, mapfile, marker_file, window_size="1 Mb", header=True, output_folder=str(tmp_path))
    assert len(res) >= 1
    g = res[0]
    assert {"WPPA", "chr", "numSNP", "estimateGenVar"} <= set(g.columns) and g["WPPA"].between(0, 1).all()
    assert list(g.sort_values("window")["numSNP"]) == [2, 1, 2]        # map.txt: chr 1 windows [0,1Mb) and [1,2Mb), chr 2 [0,1Mb)
    res = api.GWAS(model, mapfile, marker_file, window_size="1 Mb", sliding_window=True, header=True, output_folder=str(tmp_path))
    assert "WPPA" in res[0].columns
    res, props = api.GWAS(model, mapfile, marker_file, window_size="1 Mb", output_winVarProps=True, header=True, output_folder=str(tmp_path))
    assert len(res) >= 1 and len(props) >= 1 and props[0].shape == (8, 3)
    res = api.GWAS(model, mapfile, marker_file, window_size="1 Mb", threshold=0.01, header=True, output_folder=str(tmp_path))
    assert "WPPA" in res[0].columns


def test_window_genetic_correlation_matches_literal_restatement(tmp_path):
    """GWAS.jl:199-237: per sample and window cov / cor of the two traits' window genomic values."""
    X, s1, f1, mapf, chrom, pos = _case(tmp_path, seed=5)
    rng = np.random.default_rng(9)
    s2 = np.where(rng.random(s1.shape) < 0.2, rng.standard_normal(s1.shape), 0.0).astype(np.float32)
    s2[:, :5] = 0.7 * s1[:, :5]                        # some shared signal
    f2 = str(tmp_path / "MCMC_samples_marker_effects_geno_y2.txt")
    pd.DataFrame(s2, columns=[f"m{j + 1}" for j in range(s1.shape[1])]).to_csv(f2, index=False, float_format="%.9g")
    res = GWAS(X, mapf, f1, f2, window_size="1 Mb", GWAS=False, genetic_correlation=True, output_folder=str(tmp_path),
               _engine=OracleEngine("dense"))
    tab = res[-1]
    assert list(tab.columns) == ["trait", "window", "chr", "wStart", "wEnd", "start_SNP", "end_SNP", "numSNP", "estimate_cov",
                                 "std_cov", "estimate_cor", "std_cor"]
    win = build_windows(chrom, pos, 1_000_000, False)
    X64 = X.astype(np.float64)
    ns, nw = s1.shape[0], len(win["nsnp"])
    gcov, gcor = np.zeros((ns, nw)), np.zeros((ns, nw))
    for i in range(ns):
        for w in range(nw):
            a, b = win["col_start"][w], win["col_end"][w]
            b1, b2 = X64[:, a:b] @ s1[i, a:b].astype(np.float64), X64[:, a:b] @ s2[i, a:b].astype(np.float64)
            gcov[i, w] = np.cov(b1, b2)[0, 1]
            with np.errstate(divide="ignore", invalid="ignore"):
                c = np.corrcoef(b1, b2)[0, 1]
            gcor[i, w] = c if np.isfinite(c) else 0.0
    np.testing.assert_allclose(tab["estimate_cov"], gcov.mean(0), atol=1e-9)
    np.testing.assert_allclose(tab["estimate_cor"], gcor.mean(0), atol=1e-9)
    with pytest.raises(ValueError, match="exactly two"):
        GWAS(X, mapf, f1, genetic_correlation=True, _engine=OracleEngine("dense"), output_folder=str(tmp_path))


@pytest.mark.gpu
def test_device_window_sums2_match_numpy():
    import jwas_jl_amd as J
    rng = np.random.default_rng(2)
    n, p = 600, 200
    X = rng.standard_normal((n, p)).astype(np.float32)
    a1 = np.where(rng.random(p) < 0.2, rng.standard_normal(p), 0).astype(np.float32)
    a2 = np.where(rng.random(p) < 0.2, rng.standard_normal(p), 0).astype(np.float32)
    nz = np.flatnonzero((a1 != 0) | (a2 != 0))
    wptr = np.array([0, 9, 9, 20, nz.size], dtype=np.int32)
    e = J.HipEngine(0); e.load_dense(X)
    got = e.window_sums2(wptr, nz, a1[nz], a2[nz])
    o = OracleEngine("dense"); o.load_dense(X)
    want = o.window_sums2(wptr, nz, a1[nz], a2[nz])
    for g, w in zip(got, want):
        np.testing.assert_allclose(g, w, rtol=1e-11, atol=1e-9)
    e.close()
