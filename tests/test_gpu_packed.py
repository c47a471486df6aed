"""2-bit packed device storage (the reference's Packed2BitBackend kept packed in HBM): every kernel decodes on the fly.  Decoded
columns, x'x, Grams, X alpha and the window sums equal the dense fp32 path on the decoded matrix bit for bit; the sweep's block
right-hand sides are formed in the packed update role's own order (centring factored out of the sum) and are compared with the
oracle in that order -- and with the dense chain at the reference's own stream-vs-dense tolerance."""
import numpy as np
import pytest

from conftest import make_dataset
from oracle_engine import OracleEngine
from jwas_jl_amd import streaming as S

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def engines():
    import jwas_jl_amd as J
    a, b = J.HipEngine(0), J.HipEngine(0)
    yield a, b
    a.close(); b.close()


def _packed_inputs(n, p, seed, missing=True):
    d = make_dataset(n=n, p=p, ncausal=8, seed=seed, center=False)
    raw = d["raw"].astype(np.float64)
    if missing:
        rng = np.random.default_rng(seed)
        raw[rng.integers(0, n, 40), rng.integers(0, p, 40)] = 9
    return d, raw


@pytest.mark.parametrize("n,p,centered", [(403, 300, True), (256, 130, False), (1021, 77, True)])
def test_decode_xpx_gram_match_the_dense_path(engines, tmp_path, n, p, centered):
    dense, packed = engines
    d, raw = _packed_inputs(n, p, 3)
    prefix = S.prepare_streaming_genotypes(raw, tmp_path / "g", quality_control=False, center=centered)
    b = S.load_streaming_backend(prefix)
    X = S.decode_markers(b)
    packed.load_jgb2(prefix + ".jgb2")
    info = packed.storage_info()
    assert info["kind"] == "packed2bit" and info["n"] == n and info["p"] == p
    assert info["bytes"] < X.nbytes / 12
    assert np.array_equal(packed.get_columns(0, p), X)                       # decode_marker! on the device
    assert np.array_equal(packed.get_columns(p - 3, 3), X[:, p - 3:])
    dense.load_dense(X)
    for e in (dense, packed):
        e.setup_blocks(64, "f64")
    assert np.array_equal(packed.xpx(), dense.xpx())
    np.testing.assert_allclose(packed.xpx(), b["xpRinvx"], rtol=2e-5) if centered else None
    for i in range(dense.nblocks):
        assert np.array_equal(packed.gram(i), dense.gram(i))
    for e in (dense, packed):
        e.setup_blocks(64, "mfma")
    for i in range(dense.nblocks):
        assert np.array_equal(packed.gram(i), dense.gram(i))
    import jwas_jl_amd as J
    with pytest.raises(J.JwasHipError, match="2-bit packed"):
        packed.layout()
    packed.set_xpx(b["xpRinvx"])                                             # the sidecar can replace the device x'x
    np.testing.assert_array_equal(packed.xpx(), b["xpRinvx"])


@pytest.mark.parametrize("method,bs,n", [("BayesC", 64, 530), ("BayesC", 512, 2300), ("BayesR", 128, 530), ("MTBayesC", 64, 1100), ("BayesB", 256, 530)])
def test_packed_chain_matches_its_oracle_order_and_the_dense_chain(engines, method, bs, n):
    """The sweep on 2-bit packed storage (update_role_wide: 1024-row slices, one dword = 16 individuals per lane, the centring
    factored out of the sum) against (a) the oracle with the block right-hand sides in the packed path's OWN order (oracle
    dot_xr / set_packed_source): identical indicator / class trajectories, effects within 5e-6, the same number of changes in
    every sweep; (b) the dense chain on the decoded matrix: within 1e-4 of the effects' scale -- the tolerance the reference
    itself accepts between its streaming and dense paths (test/unit/test_streaming_codec.jl:100,104).  n = 2300: three
    1024-row slices, the last one ragged; missing codes present."""
    dense, packed = engines
    p = 2 * bs + 37
    d, raw = _packed_inputs(n, p, 17)
    miss = raw == 9
    codes = np.where(miss, 3, raw).astype(np.uint8)
    means = np.array([raw[~miss[:, j], j].mean(dtype=np.float32) for j in range(p)], dtype=np.float32)
    payload = S.pack_2bit(codes)
    v = np.where(miss, means[None, :], raw.astype(np.float32)).astype(np.float32)
    X = np.asfortranarray(v - means[None, :])
    packed.load_packed2bit(payload, n, means, centered=True)
    dense.load_dense(X)
    orc = OracleEngine("lookahead")
    orc.load_dense(X)
    t = 2 if method == "MTBayesC" else 1
    for e in (dense, packed, orc):
        e.setup_blocks(bs, "f64")
        e.init_state(method, t)
    y = d["y"] - d["y"].mean()
    for k in range(t):
        for e in (dense, packed, orc):
            e.set_residual((1 + k) * y, k)
    if method == "BayesR":
        for e in (dense, packed, orc):
            e.set_state(0, delta=np.ones(p, dtype=np.int32))
    vare = np.float32(0.5 * y.var())
    varg = np.float32(0.5 * y.var() / (0.1 * 0.4 * p))
    if method == "BayesC":
        kw = dict(vare=vare, var_effect=varg, pi=0.9)
    elif method == "BayesB":
        kw = dict(vare=vare, var_effect=varg, var_effect_vec=np.full(p, varg, dtype=np.float32), pi=0.8)
    elif method == "BayesR":
        kw = dict(vare=vare, var_effect=np.float32(5 * varg), pi_classes=np.array([0.9, 0.05, 0.03, 0.02]))
    else:
        kw = dict(vare=np.array([[vare, 0.1 * vare], [0.1 * vare, 2 * vare]], dtype=np.float32),
                  var_effect=np.array([[varg, 0.2 * varg], [0.2 * varg, varg]], dtype=np.float32),
                  log_prior_states=np.log(np.array([0.8, 0.05, 0.05, 0.1])))
    try:
        orc.set_packed_source(codes, means, centered=True)
        for it in range(1, 11):
            dense.sweep(iteration=it, seed=3, **kw)
            sp = packed.sweep(iteration=it, seed=3, **kw)
            so = orc.sweep(iteration=it, seed=3, **kw)
            assert so["n_events"] == sp["n_events"], f"iteration {it}"
        for k in range(t):
            ap, bp, dp = packed.get_state(k)
            ao, bo, do = orc.get_state(k)
            assert np.array_equal(do, dp)
            np.testing.assert_allclose(ap, ao, rtol=0, atol=5e-6)
            np.testing.assert_allclose(packed.get_residual(k), orc.get_residual(k), rtol=0, atol=3e-5)
            ad = dense.get_state(k)[0]
            scale = max(float(np.abs(ad).max()), 1e-3)
            np.testing.assert_allclose(ap, ad, rtol=0, atol=1e-4 * scale)
        # independent-block mode goes through the same accessor (k_indep_rhs)
        sp = packed.sweep(iteration=11, seed=3, independent_blocks=True, **kw)
        so = orc.sweep(iteration=11, seed=3, independent_blocks=True, **kw)
        assert so["n_events"] == sp["n_events"]
        np.testing.assert_allclose(packed.get_state(0)[0], orc.get_state(0)[0], rtol=0, atol=5e-6)
    finally:
        orc.set_packed_source(None, None)
    # X alpha decodes the payload: identical to the dense kernel on the decoded matrix for the same effects
    a_same = packed.get_state(0)[0]
    dense.set_state(0, alpha=a_same)
    assert np.array_equal(dense.mul_alpha(0), packed.mul_alpha(0))


def test_packed_synth_equals_dense_synth(engines):
    dense, packed = engines
    n, p = 1000, 200
    dense.alloc_dense(n, p); dense.synth(77, kind=0, center=True, marker_offset=5)
    packed.alloc_packed(n, p, centered=True); packed.synth(77, kind=0, center=True, marker_offset=5)
    assert np.array_equal(dense.get_columns(0, p), packed.get_columns(0, p))
    import jwas_jl_amd as J
    with pytest.raises(J.JwasHipError, match="0/1/2 genotypes only"):
        packed.synth(77, kind=1)


def test_stream_mode_runmcmc_on_the_device(engines, tmp_path):
    import pandas as pd
    import jwas_jl_amd.api as api
    d = make_dataset(n=300, p=700, ncausal=5, seed=12, center=False)
    ids = [str(i) for i in range(300)]
    raw = d["raw"].astype(np.float64)
    gdf = pd.DataFrame(raw, columns=[f"s{j}" for j in range(700)])
    gdf.insert(0, "ID", ids)
    ph = pd.DataFrame({"ID": ids, "y1": d["y"]})
    prefix = S.prepare_streaming_genotypes(raw, tmp_path / "st", obs_ids=ids, marker_ids=list(gdf.columns[1:]))
    outs = []
    for mode in ("dense", "stream"):
        geno = api.get_genotypes(gdf if mode == "dense" else prefix, method="BayesC", Pi=0.95, storage=mode)
        model = api.build_model("y1 = intercept + geno")
        outs.append(api.runMCMC(model, ph, chain_length=60, burnin=10, seed=4, output_folder=str(tmp_path / mode),
                                block_size=128, gram_mode="f64"))
    a, b = (o["marker effects geno"] for o in outs)
    np.testing.assert_allclose(a["Estimate"], b["Estimate"], atol=1e-4)       # test_streaming_codec.jl:100,104
    np.testing.assert_allclose(a["Model_Frequency"], b["Model_Frequency"], atol=1e-4)
    np.testing.assert_allclose(outs[0]["EBV_y1"]["EBV"], outs[1]["EBV_y1"]["EBV"], atol=1e-3)


def test_ebv_products_and_window_sums_on_packed_storage(engines, tmp_path):
    """X*alpha (sparse list kernel and the dense 16-deep kernel) and the GWAS window sums decode the packed payload on
    the fly: identical to the dense path on the decoded matrix."""
    dense, packed = engines
    d, raw = _packed_inputs(517, 400, 8)
    prefix = S.prepare_streaming_genotypes(raw, tmp_path / "g", quality_control=False, center=True)
    X = S.decode_markers(S.load_streaming_backend(prefix))
    packed.load_jgb2(prefix + ".jgb2")
    dense.load_dense(X)
    rng = np.random.default_rng(0)
    a_sparse = np.zeros(400, dtype=np.float32); a_sparse[rng.choice(400, 12, replace=False)] = rng.standard_normal(12)
    a_dense = rng.standard_normal(400).astype(np.float32)
    for e in (dense, packed):
        e.setup_blocks(64, "f64"); e.init_state("BayesC")
    for a in (a_sparse, a_dense):
        for e in (dense, packed):
            e.set_state(alpha=a)
        gd, gp = dense.mul_alpha(), packed.mul_alpha()
        assert np.array_equal(gd, gp)
        np.testing.assert_allclose(gd, X.astype(np.float64) @ a.astype(np.float64), atol=2e-4)
    nz = np.flatnonzero(a_sparse)
    wptr = np.array([0, nz.size, nz.size + 4, nz.size + 9], dtype=np.int32)
    idx = np.concatenate([nz, nz[:4], nz[4:9]]).astype(np.int32)
    sd, qd = dense.window_sums(wptr, idx, a_sparse[idx])
    sp, qp = packed.window_sums(wptr, idx, a_sparse[idx])
    assert np.array_equal(sd, sp) and np.array_equal(qd, qp)


@pytest.mark.parametrize("method,bs,m,n", [("BayesC", 64, 2, 530), ("BayesC", 64, 4, 2300), ("BayesR", 128, 4, 530), ("BayesB", 256, 2, 1100)])
def test_packed_grouped_launches_match_the_oracle(engines, method, bs, m, n):
    """GROUPED LAUNCHES on 2-bit packed storage (k_group_step<., PackedCols>: update_role_wide on the group's columns with the
    merged change list): device vs the oracle's grouped restatement with the right-hand sides in the packed order -- identical
    trajectories, effects within 5e-6; ragged last group, missing codes, a ragged last 1024-row slice (n = 2300)."""
    _, packed = engines
    p = bs * (2 * m + 1) + 37
    d, raw = _packed_inputs(n, p, 23)
    miss = raw == 9
    codes = np.where(miss, 3, raw).astype(np.uint8)
    means = np.array([raw[~miss[:, j], j].mean(dtype=np.float32) for j in range(p)], dtype=np.float32)
    v = np.where(miss, means[None, :], raw.astype(np.float32)).astype(np.float32)
    X = np.asfortranarray(v - means[None, :])
    packed.load_packed2bit(S.pack_2bit(codes), n, means, centered=True)
    orc = OracleEngine("lookahead")
    orc.load_dense(X)
    for e in (packed, orc):
        e.setup_blocks(bs, "f64")
        e.setup_groups(m, "f64")
        e.init_state(method)
        e.set_residual(d["y"] - d["y"].mean())
    if method == "BayesR":
        for e in (packed, orc):
            e.set_state(0, delta=np.ones(p, dtype=np.int32))
    y = d["y"] - d["y"].mean()
    vare = np.float32(0.5 * y.var())
    varg = np.float32(0.5 * y.var() / (0.1 * 0.4 * p))
    if method == "BayesC":
        kw = dict(vare=vare, var_effect=varg, pi=0.9)
    elif method == "BayesB":
        kw = dict(vare=vare, var_effect=varg, var_effect_vec=np.full(p, varg, dtype=np.float32), pi=0.8)
    else:
        kw = dict(vare=vare, var_effect=np.float32(5 * varg), pi_classes=np.array([0.9, 0.05, 0.03, 0.02]))
    try:
        orc.set_packed_source(codes, means, centered=True)
        moved = 0
        for it in range(1, 11):
            sp = packed.sweep(iteration=it, seed=3, group_launch=True, **kw)
            so = orc.sweep(iteration=it, seed=3, group_launch=True, **kw)
            assert so["n_events"] == sp["n_events"], f"iteration {it}"
            moved += int(sp["n_events"])
        assert moved > 20
        ap, _, dp = packed.get_state(0)
        ao, _, do = orc.get_state(0)
        assert np.array_equal(do, dp)
        np.testing.assert_allclose(ap, ao, rtol=0, atol=5e-6)
        np.testing.assert_allclose(packed.get_residual(0), orc.get_residual(0), rtol=0, atol=3e-5)
    finally:
        orc.set_packed_source(None, None)
