"""GPU parity tests: the HIP sweep (through the C ABI) against the CPU oracle on identical seeded
inputs.  Tolerances: delta / class trajectories bit-exact; alpha and the residual within a few
fp32 ulp (the stated floating-point tolerance is 1e-4 on posterior means, the reference's own
dense-vs-stream tolerance, test/unit/test_streaming_codec.jl:100,104; observed agreement is ~1e-7).
"""
import numpy as np
import pytest

from conftest import make_dataset
from oracle_engine import OracleEngine
import oracle as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hip():
    import jwas_jl_amd as J
    e = J.HipEngine(0)
    yield e
    e.close()


def _pair(hip, data, block_size, method, ntraits=1, gram_mode="f64"):
    X = data["X"]
    orc = OracleEngine(form="lookahead")
    orc.load_dense(X)
    orc.setup_blocks(block_size)
    orc.init_state(method, ntraits)
    hip.load_dense(X)
    hip.setup_blocks(block_size, gram_mode)
    hip.init_state(method, ntraits)
    return orc, hip


def _hyper(data, pi=0.95):
    vare = np.float32(0.5 * data["y"].var())
    sum2pq = float((2 * data["freq"] * (1 - data["freq"])).sum())
    varg = np.float32(0.5 * data["y"].var() / ((1 - pi) * sum2pq))
    return vare, varg


def _compare_state(orc, hip, t=0, atol=2e-6):
    ao, bo, do = orc.get_state(t)
    ah, bh, dh = hip.get_state(t)
    assert np.array_equal(do, dh), f"delta trajectories diverged at {np.flatnonzero(do != dh)[:5]}"
    np.testing.assert_allclose(ah, ao, rtol=0, atol=atol)
    np.testing.assert_allclose(bh, bo, rtol=0, atol=atol)
    np.testing.assert_allclose(hip.get_residual(t), orc.get_residual(t), rtol=0, atol=2e-5)


def test_xpx_and_gram_exact(hip, small_data):
    orc, hip = _pair(hip, small_data, 64, "BayesC")
    xo, xh = orc.xpx(), hip.xpx()
    np.testing.assert_allclose(xh, xo, rtol=1.2e-7, atol=0)
    assert (xh == xo).mean() > 0.999
    off = 0
    for i in range(hip.nblocks):
        G = hip.gram(i)
        b = G.shape[0]
        Go = orc.grams_packed()[off:off + b * b].reshape(b, b)
        off += b * b
        np.testing.assert_allclose(G, Go, rtol=1.2e-7, atol=1e-6)
        assert np.array_equal(G, G.T)


@pytest.mark.parametrize("bs", [64, 128, 256, 512, 1024])
def test_gram_mfma_close(hip, bs):
    data = make_dataset(n=1500, p=2 * bs + 37, ncausal=5, seed=5)
    orc, hip = _pair(hip, data, bs, "BayesC", gram_mode="mfma")
    off = 0
    for i in range(hip.nblocks):
        G = hip.gram(i)
        b = G.shape[0]
        Go = orc.grams_packed()[off:off + b * b].reshape(b, b)
        off += b * b
        scale = np.sqrt(np.outer(np.diag(Go), np.diag(Go)))
        assert np.abs(G - Go).max() / scale.max() < 2e-6
        assert np.array_equal(G, G.T)


@pytest.mark.parametrize("bs", [64, 128, 256, 512, 1024])
def test_bayesc_chain_parity(hip, bs):
    # n not a multiple of 256, p not a multiple of the block size
    data = make_dataset(n=777, p=3 * bs + 41, ncausal=12, seed=100 + bs)
    orc, hip = _pair(hip, data, bs, "BayesC")
    r0 = data["y"] - data["y"].mean()
    orc.set_residual(r0)
    hip.set_residual(r0)
    vare, varg = _hyper(data)
    for it in range(1, 41):
        so = orc.sweep(iteration=it, seed=2026, vare=vare, var_effect=varg, pi=0.9)
        sh = hip.sweep(iteration=it, seed=2026, vare=vare, var_effect=varg, pi=0.9)
        assert so["sum_delta"][0] == sh["sum_delta"][0], f"iteration {it}"
        assert so["n_events"] == sh["n_events"]
        np.testing.assert_allclose(sh["alpha_ss"], so["alpha_ss"], rtol=1e-6)
        np.testing.assert_allclose(sh["resid_ss"], so["resid_ss"], rtol=1e-6)
        np.testing.assert_allclose(sh["resid_sum"], so["resid_sum"], rtol=0, atol=1e-3)
    _compare_state(orc, hip)


@pytest.mark.parametrize("method,bs", [("BayesC", 512), ("BayesR", 512), ("BayesC", 256), ("BayesC", 1024)])
def test_compact_candidate_chain_and_its_fallback(hip, method, bs, capfd, monkeypatch):
    """Round 4: the compact candidate chain (sampler_st.hpp: compact_walk / compact_surprise) -- 6..64 candidates per block,
    wave 0 walks the candidates only, every other marker is verified afterwards, and a block in which a NON-candidate crosses
    its threshold ("surprise") is thrown away and re-run through the speculative rounds from its untouched entry state.
    Strong LD (near-duplicate neighbouring markers: a committed change moves its neighbour's rhs by much more than the 1/16
    candidate margin) makes surprises frequent; the chain must equal the oracle's either way, and the library's phase counters
    must show that both the compact chain and its fallback actually ran."""
    import re
    rng = np.random.default_rng(77)
    n, p = 1500, 3 * bs + 40
    base = make_dataset(n=n, p=p, ncausal=12, seed=78)
    X = base["X"].copy()
    for j in range(1, p, 2):                                   # marker j ~ marker j-1 (r^2 ~ 0.9)
        flip = rng.random(n) < 0.05
        X[:, j] = np.where(flip, X[:, j], X[:, j - 1])
    X -= X.mean(axis=0, dtype=np.float64).astype(np.float32)
    data = dict(base); data["X"] = np.asfortranarray(X.astype(np.float32))
    orc, hip = _pair(hip, data, bs, method)
    y = (data["y"] - data["y"].mean()).astype(np.float32)
    for e in (orc, hip):
        e.set_residual(y)
    monkeypatch.setenv("JWAS_HIP_DEBUG_PHASES", "1")
    vare = np.float32(0.5 * y.var())
    if method == "BayesR":
        kw = dict(vare=vare, var_effect=np.float32(0.02), pi_classes=np.array([0.9, 0.05, 0.03, 0.02]))
    else:
        kw = dict(vare=vare, var_effect=np.float32(0.004), pi=0.93)
    tried = fell = 0
    for it in range(1, 41):
        so = orc.sweep(iteration=it, seed=5, **kw)
        sh = hip.sweep(iteration=it, seed=5, **kw)
        if method == "BayesR":
            assert np.array_equal(so["class_counts"], sh["class_counts"]), f"iteration {it}"
        else:
            assert so["sum_delta"][0] == sh["sum_delta"][0], f"iteration {it}"
        m = re.findall(r"compact: blocks=(\d+) fallback=(\d+)", capfd.readouterr().err)
        assert m, "the library did not print its phase counters"
        tried += int(m[-1][0]); fell += int(m[-1][1])
    _compare_state(orc, hip, atol=5e-6)
    assert tried > (20 if bs < 1024 else 4), f"the compact chain ran in {tried} blocks only"      # (1024-marker blocks stage 31 rows at most)
    if bs <= 512 and method == "BayesC":
        assert fell > 0, "no block took the fallback: the test no longer exercises it"


@pytest.mark.parametrize("method,bs,pi", [("BayesC", 256, 0.6), ("BayesC", 512, 0.5), ("BayesC", 1024, 0.7), ("BayesC", 1024, 0.3),
                                          ("BayesR", 512, 0.5), ("BayesR", 1024, 0.6)])
def test_many_changes_per_block_parity(hip, method, bs, pi):
    """Hundreds of effect changes per block: more than the change log holds (64 entries: flush + eager application), more
    candidates than Gram rows fit in LDS (rows fetched on demand, overflow slot).  Still the sequential chain, bit for bit."""
    data = make_dataset(n=500, p=2 * bs + 37, ncausal=40, seed=900 + bs)
    orc, hip = _pair(hip, data, bs, method)
    r0 = data["y"] - data["y"].mean()
    orc.set_residual(r0)
    hip.set_residual(r0)
    vare, varg = _hyper(data, pi=pi)
    if method == "BayesR":
        kw = dict(var_effect=np.float32(0.02), pi_classes=np.array([pi, (1 - pi) * 0.5, (1 - pi) * 0.3, (1 - pi) * 0.2]))
        for e in (orc, hip):
            e.set_state(delta=np.ones(e.p, dtype=np.int32))
    else:
        kw = dict(var_effect=varg, pi=pi)
    for it in range(1, 6):
        so = orc.sweep(iteration=it, seed=13, vare=vare, **kw)
        sh = hip.sweep(iteration=it, seed=13, vare=vare, **kw)
        assert so["n_events"] == sh["n_events"], f"iteration {it}"
    assert sh["n_events"] > 0.2 * hip.p
    ao, _, do = orc.get_state()
    ah, _, dh = hip.get_state()
    assert np.array_equal(do, dh), f"indicator trajectories diverged at {np.flatnonzero(do != dh)[:5]}"
    np.testing.assert_allclose(ah, ao, rtol=0, atol=5e-6)
    np.testing.assert_allclose(hip.get_residual(), orc.get_residual(), rtol=0, atol=5e-5)


def test_bayesc_all_included_pi0(hip, small_data):
    """Pi = 0: every marker is in the model every sweep (the reference benchmark's setting,
    benchmarks/jwas_nonblock_benchmark.jl:34-51): the dense-event path."""
    orc, hip = _pair(hip, small_data, 128, "BayesC")
    r0 = small_data["y"] - small_data["y"].mean()
    orc.set_residual(r0)
    hip.set_residual(r0)
    vare, varg = np.float32(0.6), np.float32(0.002)
    for it in range(1, 6):
        so = orc.sweep(iteration=it, seed=3, vare=vare, var_effect=varg, pi=0.0)
        sh = hip.sweep(iteration=it, seed=3, vare=vare, var_effect=varg, pi=0.0)
        assert sh["sum_delta"][0] == small_data["X"].shape[1] == so["sum_delta"][0]
    _compare_state(orc, hip, atol=5e-6)


@pytest.mark.parametrize("method,bs,n,gram", [("BayesC", 512, 1100, "f64"), ("BayesC", 128, 1100, "f64"), ("BayesB", 256, 1100, "f64"),
                                              ("BayesC", 512, 5200, "mfma")])
def test_rule_d_device_against_the_literal_oracle(hip, method, bs, n, gram):
    """The device (Rule D: alpha = fmaf(c1, x, c0) under a uniform pi = 0, csrc/kernels.hpp AbcMarker::rule_d; dense_big_st on
    full 256- / 512-marker blocks, the in-lane walk on 128-marker ones) against the oracle in the reference's LITERAL operation
    order (bayesabc_update_marker!, BayesABC.jl:36-46: rhs -> gHat -> alpha, RULE_D off) on the reference benchmark's shape
    (Pi = 0, every marker in the model, benchmarks/jwas_nonblock_benchmark.jl:34-51).  Not bit for bit -- the two associate
    differently -- but within the stated floating-point tolerance: every marker included every sweep on both sides, effects
    and posterior means within 1e-4 of the effects' scale, the residual within 2e-4.  The last case is the production geometry:
    512-marker blocks through dense_big_st with the helper workgroup forming the lookahead correction, n = 5 200, MFMA Grams."""
    data = make_dataset(n=n, p=2 * bs + 77, ncausal=30, seed=4100 + bs)
    rng = np.random.default_rng(bs)
    kw = dict(vare=np.float32(0.6), var_effect=np.float32(0.002), pi=0.0)
    if method == "BayesB":
        kw["var_effect_vec"] = rng.uniform(0.001, 0.01, data["X"].shape[1]).astype(np.float32)
    r0 = data["y"] - data["y"].mean()
    try:
        O.RULE_D = False
        orc, hip = _pair(hip, data, bs, method, gram_mode=gram)
        orc.set_residual(r0); hip.set_residual(r0)
        mean_o = np.zeros(orc.p); mean_h = np.zeros(orc.p)
        nit = 20
        for it in range(1, nit + 1):
            so = orc.sweep(iteration=it, seed=17, **kw)
            sh = hip.sweep(iteration=it, seed=17, **kw)
            assert so["sum_delta"][0] == sh["sum_delta"][0] == orc.p, f"iteration {it}"
            mean_o += orc.get_state(0)[0]; mean_h += hip.get_state(0)[0]
    finally:
        O.RULE_D = True
    ao, ah = orc.get_state(0)[0], hip.get_state(0)[0]
    scale = max(float(np.abs(ao).max()), 1.0)
    assert (ao != ah).any()                  # the two orders do round differently somewhere: the comparison is not vacuous
    np.testing.assert_allclose(ah, ao, rtol=0, atol=1e-4 * scale)
    np.testing.assert_allclose(mean_h / nit, mean_o / nit, rtol=0, atol=1e-4 * scale)
    np.testing.assert_allclose(hip.get_residual(0), orc.get_residual(0), rtol=0, atol=2e-4)


@pytest.mark.parametrize("t,bs", [(3, 128), (2, 64), (3, 256)])
def test_rule_l_device_against_the_literal_oracle(hip, t, bs):
    """The device (Rule L: the linear form beta = A w + c of a marker that is and stays in the model for every trait,
    sampler_mt.hpp mt1_linear_coeffs) against the oracle in the reference's LITERAL conditional-by-conditional order
    (_MTBayesABC_samplerI!, MTBayesABC.jl:85-120; orc_set_mt_linear_form(0)) under the reference's default all-ones prior
    (config 4's regime): identical inclusion trajectories in every sweep, effects within 1e-4 of their scale.  (3, 256) is the
    dense_big_mt geometry (the sequential walk of 256-marker blocks; Rule T on the same geometry: tests/test_gpu_rule_t.py)."""
    data = make_dataset(n=900, p=4 * bs + 29, ncausal=12, seed=4200 + t)
    rng = np.random.default_rng(40 + t)
    Y = np.stack([(1 + 0.2 * k) * (data["y"] - data["y"].mean()) + 0.3 * rng.standard_normal(len(data["y"])).astype(np.float32)
                  for k in range(t)]).astype(np.float32)
    A = rng.standard_normal((t, t)); B = rng.standard_normal((t, t))
    vare = ((A @ A.T / t + np.eye(t)) * 0.5).astype(np.float32)
    varg = ((B @ B.T / t + np.eye(t)) * 0.003).astype(np.float32)
    prior = np.full(1 << t, 1e-3); prior[-1] = 1.0; prior /= prior.sum()              # (nearly) every marker in the model
    try:
        O.lib().orc_set_mt_linear_form(0)
        orc, hip = _pair(hip, data, bs, "MTBayesC", ntraits=t)
        for k in range(t):
            orc.set_residual(Y[k], k); hip.set_residual(Y[k], k)
            ones = np.ones(orc.p, dtype=np.float32)
            orc.set_state(k, delta=ones); hip.set_state(k, delta=ones)
        for it in range(1, 13):
            so = orc.sweep(iteration=it, seed=23, vare=vare, var_effect=varg, log_prior_states=np.log(prior))
            sh = hip.sweep(iteration=it, seed=23, vare=vare, var_effect=varg, log_prior_states=np.log(prior))
            assert np.array_equal(so["state_counts"], sh["state_counts"]), f"iteration {it}"
            for k in range(t):
                assert np.array_equal(orc.get_state(k)[2], hip.get_state(k)[2]), f"iteration {it}, trait {k}"
    finally:
        O.lib().orc_set_mt_linear_form(1)
    differ = 0
    for k in range(t):
        ao, ah = orc.get_state(k)[0], hip.get_state(k)[0]
        scale = max(float(np.abs(ao).max()), 1e-3)
        np.testing.assert_allclose(ah, ao, rtol=0, atol=1e-4 * scale)
        differ += int((ao != ah).sum())
    assert differ > 0                        # the linear form did apply on the device


def test_bayesc_pi_vector_and_bayesb(hip, small_data):
    p = small_data["X"].shape[1]
    rng = np.random.default_rng(0)
    pi_vec = rng.uniform(0.5, 0.99, size=p)
    var_vec = rng.uniform(0.001, 0.01, size=p).astype(np.float32)
    r0 = small_data["y"] - small_data["y"].mean()
    for method in ("BayesC", "BayesB"):
        orc, hip_ = _pair(hip, small_data, 64, method)
        orc.set_residual(r0)
        hip_.set_residual(r0)
        kw = dict(seed=9, vare=np.float32(0.5), var_effect=np.float32(0.004), pi=pi_vec)
        if method == "BayesB":
            kw["var_effect_vec"] = var_vec
        for it in range(1, 16):
            orc.sweep(iteration=it, **kw)
            hip_.sweep(iteration=it, **kw)
        _compare_state(orc, hip_)
    with pytest.raises(ValueError, match="length"):
        hip.sweep(iteration=1, seed=1, vare=0.5, var_effect=0.004, pi=np.array([0.5]))


@pytest.mark.parametrize("bs", [64, 256, 1024])
def test_bayesr_chain_parity(hip, bs):
    data = make_dataset(n=600, p=2 * bs + 19, ncausal=10, seed=300 + bs)
    orc, hip = _pair(hip, data, bs, "BayesR")
    r0 = data["y"] - data["y"].mean()
    orc.set_residual(r0)
    hip.set_residual(r0)
    orc.set_state(delta=np.ones(orc.p, dtype=np.int32))
    hip.set_state(delta=np.ones(hip.p, dtype=np.int32))
    pi4 = np.array([0.95, 0.03, 0.015, 0.005])
    vare, sig = np.float32(0.5), np.float32(0.05)
    for it in range(1, 31):
        so = orc.sweep(iteration=it, seed=77, vare=vare, var_effect=sig, pi_classes=pi4)
        sh = hip.sweep(iteration=it, seed=77, vare=vare, var_effect=sig, pi_classes=pi4)
        assert np.array_equal(so["class_counts"], sh["class_counts"]), f"iteration {it}"
        assert so["bayesr_nnz"] == sh["bayesr_nnz"]
        np.testing.assert_allclose(sh["bayesr_ssq"], so["bayesr_ssq"], rtol=1e-5)
    ao, _, do = orc.get_state()
    ah, _, dh = hip.get_state()
    assert np.array_equal(do, dh)
    np.testing.assert_allclose(ah, ao, rtol=0, atol=2e-6)
    np.testing.assert_allclose(hip.get_residual(), orc.get_residual(), rtol=0, atol=2e-5)


def test_bayesr_degenerate_priors_kat(hip):
    """test/unit/test_annotated_bayesr.jl:247-266: per-SNP priors that force delta == [2, 1]."""
    X = np.asfortranarray(np.array([[0, 2], [1, 1], [2, 0], [1, 1]], dtype=np.float32))
    hip.load_dense(X)
    hip.setup_blocks(64, "f64")
    hip.init_state("BayesR")
    hip.set_residual(np.array([0.8, -0.1, 0.3, 0.5], dtype=np.float32))
    snp_pi = np.array([[0.0, 1.0, 0.0, 0.0], [1.0, 0.0, 0.0, 0.0]])
    for seed in (1, 2, 20260327):
        hip.set_state(alpha=np.zeros(2), delta=np.ones(2, dtype=np.int32))
        hip.sweep(iteration=1, seed=seed, vare=1.0, var_effect=0.2, pi_matrix=snp_pi)
        assert hip.get_state()[2].tolist() == [2, 1]


def test_bayesr_error_contracts(hip, small_data):
    hip.load_dense(small_data["X"])
    hip.setup_blocks(64, "f64")
    hip.init_state("BayesR")
    import jwas_jl_amd as J
    with pytest.raises(J.JwasHipError, match="sum to 1"):
        hip.sweep(iteration=1, seed=1, vare=1.0, var_effect=0.2, pi_classes=[0.5, 0.1, 0.1, 0.1])
    with pytest.raises(J.JwasHipError, match="nonnegative"):
        hip.sweep(iteration=1, seed=1, vare=1.0, var_effect=0.2, pi_classes=[1.2, -0.2, 0.0, 0.0])
    with pytest.raises(J.JwasHipError, match="sigmaSq"):
        hip.sweep(iteration=1, seed=1, vare=1.0, var_effect=0.0, pi_classes=[0.95, 0.03, 0.015, 0.005])
    with pytest.raises(ValueError, match="mixture classes"):
        hip.sweep(iteration=1, seed=1, vare=1.0, var_effect=0.2, pi_classes=[0.5, 0.5])
    with pytest.raises(ValueError, match="one row per marker"):
        hip.sweep(iteration=1, seed=1, vare=1.0, var_effect=0.2, pi_matrix=np.ones((3, 4)) / 4)


@pytest.mark.parametrize("method", ["BayesC", "BayesR"])
@pytest.mark.parametrize("nreps", [3, 0])
def test_block_repetitions_parity(hip, nreps, method):
    """fast_blocks semantics: nreps within-block passes (0 = block size; BayesABC.jl:153, BayesR.jl:146-147)."""
    data = make_dataset(n=400, p=64 * 3 + 5, ncausal=6, seed=41)
    orc, hip = _pair(hip, data, 64, method)
    r0 = data["y"] - data["y"].mean()
    orc.set_residual(r0)
    hip.set_residual(r0)
    vare, varg = _hyper(data)
    kw = dict(var_effect=np.float32(20 * varg), pi_classes=np.array([0.9, 0.06, 0.03, 0.01])) if method == "BayesR" else dict(var_effect=varg, pi=0.9)
    if method == "BayesR":
        for e in (orc, hip):
            e.set_state(delta=np.ones(e.p, dtype=np.int32))
    for it in range(1, 4):
        orc.sweep(iteration=it, seed=5, vare=vare, nreps=nreps, **kw)
        hip.sweep(iteration=it, seed=5, vare=vare, nreps=nreps, **kw)
    if method == "BayesR":                       # (BayesR keeps no beta)
        ao, _, do = orc.get_state()
        ah, _, dh = hip.get_state()
        assert np.array_equal(do, dh) and (do > 1).sum() > 3
        np.testing.assert_allclose(ah, ao, rtol=0, atol=5e-6)
        np.testing.assert_allclose(hip.get_residual(), orc.get_residual(), rtol=0, atol=2e-5)
    else:
        _compare_state(orc, hip, atol=5e-6)


# (3, 512): draws parked in LDS; (3, 1024), (4, 1024): too big to park -- the serial wave reads them from HBM
@pytest.mark.parametrize("t,bs", [(2, 64), (3, 128), (3, 256), (3, 512), (3, 1024), (4, 1024)])
def test_mt_sampler1_parity(hip, t, bs):
    data = make_dataset(n=500, p=2 * bs + 13, ncausal=10, seed=500 + t)
    orc, hip = _pair(hip, data, bs, "MTBayesC", ntraits=t)
    rng = np.random.default_rng(t)
    Y = np.stack([data["y"] - data["y"].mean() + 0.3 * rng.standard_normal(len(data["y"])).astype(np.float32)
                  for _ in range(t)]).astype(np.float32)
    for k in range(t):
        orc.set_residual(Y[k], k)
        hip.set_residual(Y[k], k)
        ones = np.ones(orc.p, dtype=np.float32)
        orc.set_state(k, delta=ones)
        hip.set_state(k, delta=ones)
    A = rng.standard_normal((t, t))
    vare = (A @ A.T / t + np.eye(t)).astype(np.float32) * 0.5
    B = rng.standard_normal((t, t))
    varg = ((B @ B.T / t + np.eye(t)) * 0.002).astype(np.float32)
    prior = rng.dirichlet(np.ones(1 << t))
    lp = np.log(prior)
    for it in range(1, 21):
        so = orc.sweep(iteration=it, seed=11, vare=vare, var_effect=varg, log_prior_states=lp)
        sh = hip.sweep(iteration=it, seed=11, vare=vare, var_effect=varg, log_prior_states=lp)
        assert np.array_equal(so["state_counts"], sh["state_counts"]), f"iteration {it}"
        np.testing.assert_allclose(sh["beta_ss"], so["beta_ss"], rtol=1e-5)
        np.testing.assert_allclose(sh["resid_ss"], so["resid_ss"], rtol=1e-5, atol=1e-6 * float(np.max(np.abs(so["resid_ss"]))))
    for k in range(t):
        _compare_state(orc, hip, k, atol=5e-6)


# SKIP AND VERIFY (sampler_role_mt): sparse multi-trait blocks of 256 / 512 markers -- the serial wave walks only the 64-marker
# sub-blocks that hold a candidate, the helper waves evaluate the others; the chain must be the oracle's, and the helpers must
# have run (sweep counter 31).  `dup`: copies of the causal columns a few sub-blocks behind the originals -- when a causal marker's
# effect changes, its copy's right-hand side moves with it and the copy enters the model from a skipped sub-block (counter 30: the
# serial wave took the chain over again from there).
@pytest.mark.parametrize("method,t,bs,dup", [("MTBayesC", 3, 512, False), ("MTBayesC", 3, 256, False), ("MTBayesC", 2, 512, True),
                                             ("MTBayesC", 3, 512, True), ("MTBayesC_II", 3, 512, True), ("MTBayesC_II", 2, 256, False),
                                             ("MegaBayesC", 3, 512, True), ("MTBayesB", 3, 512, False),
                                             # 1024-marker blocks: 16 sub-blocks, up to three per helper wave; four traits: draws read from HBM
                                             ("MTBayesC", 3, 1024, True), ("MTBayesC_II", 2, 1024, False), ("MTBayesC", 4, 1024, False)])
def test_mt_skip_and_verify_sparse_blocks(hip, method, t, bs, dup):
    data = make_dataset(n=600, p=3 * bs + 150, ncausal=24, h2=0.7, seed=910 + t + bs)
    X = data["X"].copy()
    if dup:
        for j in data["causal"]:
            for off in (64, 130, 200):
                k = j + off
                if k // bs == j // bs and k < X.shape[1] and k not in data["causal"]:      # (same block, a later sub-block)
                    X[:, k] = X[:, j]
        data = dict(data, X=np.asfortranarray(X))
    orc, hip = _pair(hip, data, bs, method, ntraits=t)
    rng = np.random.default_rng(17 + t)
    Y = np.stack([data["y"] - data["y"].mean() + 0.3 * rng.standard_normal(len(data["y"])).astype(np.float32)
                  for _ in range(t)]).astype(np.float32)
    for k in range(t):
        orc.set_residual(Y[k], k)
        hip.set_residual(Y[k], k)
    A = rng.standard_normal((t, t))
    vare = (A @ A.T / t + np.eye(t)).astype(np.float32) * 0.5
    B = rng.standard_normal((t, t))
    varg = ((B @ B.T / t + np.eye(t)) * 0.01).astype(np.float32)
    kw = dict(vare=vare, var_effect=varg)
    if method == "MegaBayesC":
        kw = dict(vare=np.diag(np.diag(vare)).astype(np.float32), var_effect=np.diag(np.diag(varg)).astype(np.float32), pi=np.full(t, 0.998))
    else:
        prior = np.full(1 << t, 0.03 / ((1 << t) - 1))
        prior[0] = 0.97
        kw["log_prior_states"] = np.log(prior)
        if method == "MTBayesB":                     # a covariance per marker (the helper waves use the marker's own constants)
            Bm = rng.standard_normal((orc.p, t, t))
            kw["var_effect_matrix"] = ((Bm @ Bm.transpose(0, 2, 1) / t + np.eye(t)) * 0.01).astype(np.float32)
    helped, taken_over = 0, 0
    for it in range(1, 31):
        so = orc.sweep(iteration=it, seed=23, **kw)
        sh = hip.sweep(iteration=it, seed=23, **kw)
        assert so["n_events"] == sh["n_events"], f"iteration {it}"
        c = hip.last_sweep_counters()
        helped += int(c[31]); taken_over += int(c[30])
    for k in range(t):
        _compare_state(orc, hip, k, atol=5e-6)
    assert helped > 20, f"the helper waves evaluated only {helped} sub-blocks"
    if dup and method in ("MTBayesC", "MTBayesC_II"):
        assert taken_over > 0, "no skipped marker ever moved: the take-over path did not run"


def test_mt_three_resident_block_sizes_mid_chain(hip):
    """mcmc.pick_block_size_mt's three levels (256 while the chain is dense, 512 once sparse, 1024 below 0.5 % turnover): a sparse
    three-trait sampler-I chain whose host switches among the three resident block sizes between sweeps is the same chain as the
    oracle's, which follows the same switches -- the draws do not depend on the partition."""
    data = make_dataset(n=500, p=2 * 1024 + 200, ncausal=16, h2=0.7, seed=33)
    t = 3
    orc, hip = _pair(hip, data, 256, "MTBayesC", ntraits=t)
    for bs in (512, 1024):
        orc.add_block_size(bs); hip.add_block_size(bs, "f64")
    rng = np.random.default_rng(3)
    Y = np.stack([data["y"] - data["y"].mean() + 0.3 * rng.standard_normal(len(data["y"])).astype(np.float32) for _ in range(t)]).astype(np.float32)
    for k in range(t):
        orc.set_residual(Y[k], k); hip.set_residual(Y[k], k)
    A = rng.standard_normal((t, t)); B = rng.standard_normal((t, t))
    vare = (A @ A.T / t + np.eye(t)).astype(np.float32) * 0.5
    varg = ((B @ B.T / t + np.eye(t)) * 0.01).astype(np.float32)
    prior = np.full(1 << t, 0.03 / ((1 << t) - 1)); prior[0] = 0.97
    kw = dict(vare=vare, var_effect=varg, log_prior_states=np.log(prior))
    helped = 0
    for it in range(1, 25):
        bs = (256, 512, 1024, 1024, 512, 1024)[(it // 2) % 6]
        orc.select_block_size(bs); hip.select_block_size(bs)
        so = orc.sweep(iteration=it, seed=29, **kw)
        sh = hip.sweep(iteration=it, seed=29, **kw)
        assert so["n_events"] == sh["n_events"], f"iteration {it} (block size {bs})"
        helped += int(hip.last_sweep_counters()[31])
    for k in range(t):
        _compare_state(orc, hip, k, atol=5e-6)
    assert helped > 20


# t <= 3: per-marker state tables (k_prepare_mt2); t = 4: states evaluated on the fly
@pytest.mark.parametrize("t,bs,nreps", [(2, 64, 1), (2, 128, 1), (3, 128, 1), (4, 64, 1), (2, 64, 2), (3, 512, 1), (3, 1024, 1), (4, 256, 1), (3, 64, 2)])
def test_mt_sampler2_parity(hip, t, bs, nreps):
    """Joint-state Gibbs sampler II (MTBayesABC.jl:129-210) against the oracle's restatement."""
    data = make_dataset(n=500, p=2 * bs + 13, ncausal=10, seed=600 + t)
    orc, hip = _pair(hip, data, bs, "MTBayesC_II", ntraits=t)
    rng = np.random.default_rng(10 + t)
    Y = np.stack([data["y"] - data["y"].mean() + 0.3 * rng.standard_normal(len(data["y"])).astype(np.float32)
                  for _ in range(t)]).astype(np.float32)
    for k in range(t):
        orc.set_residual(Y[k], k)
        hip.set_residual(Y[k], k)
    A = rng.standard_normal((t, t))
    vare = (A @ A.T / t + np.eye(t)).astype(np.float32) * 0.5
    B = rng.standard_normal((t, t))
    varg = ((B @ B.T / t + np.eye(t)) * 0.002).astype(np.float32)
    prior = rng.dirichlet(np.ones(1 << t)) * 0.2
    prior[0] += 0.8                                     # most markers out of the model
    lp = np.log(prior)
    for it in range(1, 13):
        so = orc.sweep(iteration=it, seed=13, vare=vare, var_effect=varg, log_prior_states=lp, nreps=nreps)
        sh = hip.sweep(iteration=it, seed=13, vare=vare, var_effect=varg, log_prior_states=lp, nreps=nreps)
        assert np.array_equal(so["state_counts"], sh["state_counts"]), f"iteration {it}"
        np.testing.assert_allclose(sh["beta_ss"], so["beta_ss"], rtol=1e-5)
        np.testing.assert_allclose(sh["resid_ss"], so["resid_ss"], rtol=1e-5, atol=1e-6 * float(np.max(np.abs(so["resid_ss"]))))
    assert so["state_counts"][1:].sum() > 0
    for k in range(t):
        _compare_state(orc, hip, k, atol=5e-6)


def test_mt_sampler2_restrictive_prior_all_or_none(hip):
    """Sampler II exists for priors with zero-probability states ('a locus affects all traits or none',
    MTBayesABC.jl:4): only states 00 and 11 may appear; an all-zero prior is the reference's error (:190)."""
    data = make_dataset(n=300, p=150, ncausal=5, seed=71)
    orc, hip = _pair(hip, data, 64, "MTBayesC_II", ntraits=2)
    y = data["y"] - data["y"].mean()
    for k in range(2):
        orc.set_residual(y * (1 + k), k)
        hip.set_residual(y * (1 + k), k)
    vare = np.array([[0.5, 0.1], [0.1, 0.9]], dtype=np.float32)
    varg = np.array([[0.003, 0.001], [0.001, 0.004]], dtype=np.float32)
    with np.errstate(divide="ignore"):
        lp = np.log(np.array([0.7, 0.0, 0.0, 0.3]))
    for it in range(1, 9):
        so = orc.sweep(iteration=it, seed=2, vare=vare, var_effect=varg, log_prior_states=lp)
        sh = hip.sweep(iteration=it, seed=2, vare=vare, var_effect=varg, log_prior_states=lp)
        assert sh["state_counts"][1] == 0 and sh["state_counts"][2] == 0
        assert np.array_equal(so["state_counts"], sh["state_counts"])
    assert sh["state_counts"][3] > 0
    for k in range(2):
        _compare_state(orc, hip, k, atol=5e-6)
    import jwas_jl_amd as J
    with pytest.raises(J.JwasHipError, match="state probabilities are zero or invalid"):
        hip.sweep(iteration=1, seed=2, vare=vare, var_effect=varg, log_prior_states=np.full(4, -np.inf))


@pytest.mark.parametrize("t,bs", [(2, 64), (3, 256), (4, 128), (3, 1024)])
def test_mega_bayesc_parity(hip, t, bs):
    """megaBayesABC! (BayesABC.jl:1-8): t independent single-trait chains in one pass over X; trait 0 is
    bit-identical to the single-trait BayesC chain (same draw slot)."""
    data = make_dataset(n=500, p=2 * bs + 29, ncausal=10, seed=700 + t)
    orc, hip = _pair(hip, data, bs, "MegaBayesC", ntraits=t)
    rng = np.random.default_rng(20 + t)
    Y = np.stack([(1 + 0.5 * k) * (data["y"] - data["y"].mean()) + 0.3 * rng.standard_normal(len(data["y"])).astype(np.float32)
                  for k in range(t)]).astype(np.float32)
    for k in range(t):
        orc.set_residual(Y[k], k)
        hip.set_residual(Y[k], k)
    v, g = _hyper(data)
    vare = np.diag([float(v) * (1 + 0.3 * k) for k in range(t)]).astype(np.float32)
    varg = np.diag([float(g) * (1 + 0.2 * k) for k in range(t)]).astype(np.float32)
    pis = np.array([0.95 - 0.1 * k for k in range(t)])
    for it in range(1, 16):
        so = orc.sweep(iteration=it, seed=17, vare=vare, var_effect=varg, pi=pis)
        sh = hip.sweep(iteration=it, seed=17, vare=vare, var_effect=varg, pi=pis)
        assert np.array_equal(so["sum_delta"], sh["sum_delta"]), f"iteration {it}"
    for k in range(t):
        _compare_state(orc, hip, k, atol=5e-6)
    # trait 0 == the single-trait chain
    a_mega, b_mega, d_mega = hip.get_state(0)
    r_mega = hip.get_residual(0)
    hip.init_state("BayesC", 1)
    hip.set_residual(Y[0])
    for it in range(1, 16):
        hip.sweep(iteration=it, seed=17, vare=vare[0, 0], var_effect=varg[0, 0], pi=float(pis[0]))
    a1, b1, d1 = hip.get_state(0)
    assert np.array_equal(d1, d_mega)
    np.testing.assert_allclose(a1, a_mega, rtol=0, atol=5e-6)
    np.testing.assert_allclose(hip.get_residual(0), r_mega, rtol=0, atol=2e-5)


@pytest.mark.parametrize("method,bs,nreps", [("BayesC", 64, 0), ("BayesC", 256, 1), ("BayesR", 128, 2),
                                             ("MTBayesC", 64, 1), ("MTBayesC_II", 64, 1), ("MegaBayesC", 128, 0)])
def test_independent_blocks_parity(hip, method, bs, nreps):
    """independent_blocks=true (BayesABC.jl:190-255, BayesR.jl:195-273, MTBayesABC.jl:335-440): all blocks sampled
    concurrently from the same residual snapshot, reconcile afterwards -- against the oracle's restatement."""
    data = make_dataset(n=450, p=3 * bs + 21, ncausal=8, seed=900 + bs)
    t = 1 if method in ("BayesC", "BayesR") else 2
    X = data["X"]
    orc = OracleEngine(form="block")
    orc.load_dense(X)
    orc.setup_blocks(bs)
    orc.init_state(method, t)
    hip.load_dense(X)
    hip.setup_blocks(bs, "f64")
    hip.init_state(method, t)
    y = data["y"] - data["y"].mean()
    for k in range(t):
        orc.set_residual((1 + 0.5 * k) * y, k)
        hip.set_residual((1 + 0.5 * k) * y, k)
    if method == "BayesR":
        ones = np.ones(orc.p, dtype=np.int32)
        orc.set_state(0, delta=ones); hip.set_state(0, delta=ones)
    vare, varg = _hyper(data)
    if method == "BayesC":
        kw = dict(vare=vare, var_effect=varg, pi=0.9)
    elif method == "BayesR":
        kw = dict(vare=vare, var_effect=np.float32(varg * 5), pi_classes=np.array([0.9, 0.05, 0.03, 0.02]))
    elif method == "MegaBayesC":
        kw = dict(vare=np.diag([vare, 2 * vare]).astype(np.float32), var_effect=np.diag([varg, varg]).astype(np.float32),
                  pi=np.array([0.9, 0.8]))
    else:
        kw = dict(vare=np.array([[vare, 0.1 * vare], [0.1 * vare, 2 * vare]], dtype=np.float32),
                  var_effect=np.array([[varg, 0.2 * varg], [0.2 * varg, varg]], dtype=np.float32),
                  log_prior_states=np.log(np.array([0.8, 0.05, 0.05, 0.1])))
    for it in range(1, 9):
        so = orc.sweep(iteration=it, seed=23, nreps=nreps, independent_blocks=True, **kw)
        sh = hip.sweep(iteration=it, seed=23, nreps=nreps, independent_blocks=True, **kw)
        assert so["n_events"] == sh["n_events"] or nreps != 1, f"iteration {it}"
        np.testing.assert_allclose(sh["resid_ss"], so["resid_ss"], rtol=1e-5, atol=1e-6 * float(np.max(np.abs(so["resid_ss"]))))
    for k in range(t):
        _compare_state(orc, hip, k, atol=5e-6)
    # the device's lookahead chain on the same inputs is a different (the exact) chain
    assert sh["n_events"] > 0


def test_real_valued_imputed_genotypes_parity(hip):
    """Single-step input (SSBR.jl:83-142): the non-genotyped rows of X are real-valued parent-average-like mixtures,
    not 0/1/2 -- the device path is the dense fp32 one, nothing assumes integer codes."""
    data = make_dataset(n=420, p=300, ncausal=8, seed=77)
    rng = np.random.default_rng(7)
    X = data["X"].copy()
    W = rng.dirichlet(np.ones(4), size=200).astype(np.float32)              # 200 "imputed" rows = mixtures of genotyped rows
    X[220:] = W @ X[rng.integers(0, 220, 4)]
    X = np.asfortranarray(X - X.mean(axis=0, dtype=np.float32))
    d2 = dict(data, X=X)
    orc, hip = _pair(hip, d2, 128, "BayesC")
    r0 = data["y"] - data["y"].mean()
    orc.set_residual(r0); hip.set_residual(r0)
    vare, varg = _hyper(data, 0.9)
    for it in range(1, 21):
        orc.sweep(iteration=it, seed=8, vare=vare, var_effect=varg, pi=0.9)
        hip.sweep(iteration=it, seed=8, vare=vare, var_effect=varg, pi=0.9)
    _compare_state(orc, hip, atol=5e-6)


def test_switching_resident_block_sizes_mid_chain(hip):
    """Two block sizes resident (jwas_hip_add_block_size); the host switches between sweeps.  The draws do not depend
    on the block size, so this is the same chain; the oracle follows the same switches."""
    data = make_dataset(n=400, p=2 * 512 + 100, ncausal=10, seed=91)
    orc, hip = _pair(hip, data, 256, "BayesC")
    orc.add_block_size(512); hip.add_block_size(512, "f64")
    import jwas_jl_amd as J
    with pytest.raises(J.JwasHipError, match="already resident"):
        hip.add_block_size(512, "f64")
    with pytest.raises(J.JwasHipError, match="not resident"):
        hip.select_block_size(128)
    r0 = data["y"] - data["y"].mean()
    orc.set_residual(r0); hip.set_residual(r0)
    vare, varg = _hyper(data, 0.9)
    for it in range(1, 19):
        bs = 512 if (it // 3) % 2 else 256
        orc.select_block_size(bs); hip.select_block_size(bs)
        assert hip.nblocks == orc.nblocks
        so = orc.sweep(iteration=it, seed=6, vare=vare, var_effect=varg, pi=0.9)
        sh = hip.sweep(iteration=it, seed=6, vare=vare, var_effect=varg, pi=0.9)
        assert so["n_events"] == sh["n_events"]
    _compare_state(orc, hip, atol=5e-6)
    # an independent-block sweep after a switch sizes its buffers for the selected configuration
    for e in (orc, hip):
        e.select_block_size(256)
    orc.sweep(iteration=19, seed=6, vare=vare, var_effect=varg, pi=0.9, independent_blocks=True)
    hip.sweep(iteration=19, seed=6, vare=vare, var_effect=varg, pi=0.9, independent_blocks=True)
    _compare_state(orc, hip, atol=5e-6)


@pytest.mark.parametrize("method,bs,gram", [("BayesC", 128, "f64"), ("BayesR", 64, "f64"), ("MTBayesC", 64, "f64"), ("BayesC", 256, "mfma")])
def test_residual_weights_parity(hip, method, bs, gram):
    """Non-unit residual weights R^-1 (mme.invweights): x'R^-1 x, X_b'R^-1 X_b, X_b'R^-1 r, r'R^-1 r
    (tools4genotypes.jl:28-31,59-78,263-266; variance_components.jl:82-98) against the oracle."""
    data = make_dataset(n=410, p=2 * bs + 17, ncausal=8, seed=123)
    rng = np.random.default_rng(5)
    rinv = (1.0 / rng.uniform(0.5, 3.0, size=410)).astype(np.float32)
    t = 2 if method == "MTBayesC" else 1
    orc = OracleEngine("lookahead")
    for e in (orc, hip):
        e.load_dense(data["X"])
        e.set_weights(rinv)
        e.setup_blocks(bs, gram)
        e.init_state(method, t)
    xo, xh = orc.xpx(), hip.xpx()
    np.testing.assert_allclose(xh, xo, rtol=2e-7)
    np.testing.assert_allclose(xo, ((data["X"].astype(np.float64) ** 2) * rinv[:, None]).sum(axis=0), rtol=1e-5)
    off = 0
    for i in range(hip.nblocks):
        G = hip.gram(i)
        b = G.shape[0]
        Go = orc.grams_packed()[off:off + b * b].reshape(b, b)
        off += b * b
        if gram == "f64":
            assert np.array_equal(G, Go)
        else:
            np.testing.assert_allclose(G, Go, rtol=0, atol=2e-4 * np.abs(Go).max())
    if gram == "mfma":
        hip.set_grams_from(orc) if hasattr(hip, "set_grams_from") else None
        return
    y = data["y"] - data["y"].mean()
    for k in range(t):
        orc.set_residual((1 + 0.4 * k) * y, k); hip.set_residual((1 + 0.4 * k) * y, k)
    if method == "BayesR":
        ones = np.ones(orc.p, dtype=np.int32)
        orc.set_state(0, delta=ones); hip.set_state(0, delta=ones)
    vare, varg = _hyper(data, 0.9)
    if method == "BayesC":
        kw = dict(vare=vare, var_effect=varg, pi=0.9)
    elif method == "BayesR":
        kw = dict(vare=vare, var_effect=np.float32(5 * varg), pi_classes=np.array([0.9, 0.05, 0.03, 0.02]))
    else:
        kw = dict(vare=np.array([[vare, 0.1 * vare], [0.1 * vare, 2 * vare]], dtype=np.float32),
                  var_effect=np.array([[varg, 0.2 * varg], [0.2 * varg, varg]], dtype=np.float32),
                  log_prior_states=np.log(np.array([0.8, 0.05, 0.05, 0.1])))
    for it in range(1, 13):
        so = orc.sweep(iteration=it, seed=31, **kw)
        sh = hip.sweep(iteration=it, seed=31, **kw)
        assert so["n_events"] == sh["n_events"]
        np.testing.assert_allclose(sh["resid_ss"], so["resid_ss"], rtol=1e-5, atol=1e-6 * float(np.max(np.abs(so["resid_ss"]))))          # r'R^-1 r
        np.testing.assert_allclose(sh["resid_sum"], so["resid_sum"], rtol=1e-4, atol=1e-3)
    for k in range(t):
        _compare_state(orc, hip, k, atol=5e-6)
    so = orc.sweep(iteration=13, seed=31, independent_blocks=True, **kw)
    sh = hip.sweep(iteration=13, seed=31, independent_blocks=True, **kw)
    _compare_state(orc, hip, 0, atol=5e-6)
    import jwas_jl_amd as J
    with pytest.raises(J.JwasHipError, match="positive and finite"):
        hip.set_weights(np.zeros(410, dtype=np.float32))
    hip.set_weights(None)                                   # back to unit weights: block configurations are dropped
    with pytest.raises(J.JwasHipError, match="setup_blocks has not been called"):
        hip.sweep(iteration=1, seed=1, **kw)


@pytest.mark.parametrize("method,t,bs,zero", [("MTBayesC", 3, 128, False), ("MTBayesC", 2, 64, False), ("MTBayesC_II", 2, 128, False),
                                              ("MTBayesC", 3, 256, False), ("MTBayesC", 2, 256, True), ("MTBayesC", 4, 256, False),      # (dense_big_mt)
                                              ("MegaBayesC", 3, 64, False), ("MTBayesC", 3, 128, True), ("MTBayesC", 2, 64, True),
                                              ("MTBayesC", 4, 128, True), ("MTBayesC_II", 2, 128, True)])
def test_multitrait_dense_blocks_parity(hip, method, t, bs, zero):
    """The default multi-trait prior puts all mass on the all-ones state (tools4genotypes.jl:357-373): every marker stays
    in the model and every block is dense -- the sampler walks such blocks sequentially; results must not change.
    zero: the other states have probability exactly 0 (log prior -inf, the reference's literal default) -- sampler I then
    takes the walk that skips the decision; else 1e-12 (the general dense walk)."""
    data = make_dataset(n=380, p=2 * bs + 41, ncausal=10, seed=800 + t)
    orc, hip = _pair(hip, data, bs, method, ntraits=t)
    rng = np.random.default_rng(t)
    y = data["y"] - data["y"].mean()
    for k in range(t):
        yk = ((1 + 0.3 * k) * y + 0.2 * rng.standard_normal(len(y))).astype(np.float32)
        orc.set_residual(yk, k); hip.set_residual(yk, k)
        ones = np.ones(orc.p, dtype=np.float32)
        a0 = (0.01 * rng.standard_normal(orc.p)).astype(np.float32)          # every marker already in the model
        orc.set_state(k, alpha=a0, beta=a0, delta=ones); hip.set_state(k, alpha=a0, beta=a0, delta=ones)
        orc.sub_xalpha(k); hip.sub_xalpha(k)
    A = rng.standard_normal((t, t))
    vare = (A @ A.T / t + np.eye(t)).astype(np.float32) * 0.5
    Bm = rng.standard_normal((t, t))
    varg = ((Bm @ Bm.T / t + np.eye(t)) * 0.002).astype(np.float32)
    if method == "MegaBayesC":
        kw = dict(vare=np.diag(np.diag(vare)), var_effect=np.diag(np.diag(varg)), pi=np.zeros(t))
    else:
        prior = np.full(1 << t, 0.0 if zero else 1e-12); prior[-1] = 1.0; prior /= prior.sum()
        with np.errstate(divide="ignore"):
            kw = dict(vare=vare, var_effect=varg, log_prior_states=np.log(prior))
    for it in range(1, 9):
        so = orc.sweep(iteration=it, seed=21, **kw)
        sh = hip.sweep(iteration=it, seed=21, **kw)
        assert so["n_events"] == sh["n_events"] == orc.p, f"iteration {it}"
        np.testing.assert_allclose(sh["resid_ss"], so["resid_ss"], rtol=1e-5, atol=1e-6 * float(np.max(np.abs(so["resid_ss"]))))
    for k in range(t):
        _compare_state(orc, hip, k, atol=5e-6)


def test_accumulate_mul_alpha_sub_xalpha(hip, small_data):
    orc, hip = _pair(hip, small_data, 64, "BayesC")
    rng = np.random.default_rng(3)
    a0 = np.where(rng.random(orc.p) < 0.1, rng.standard_normal(orc.p) * 0.05, 0).astype(np.float32)
    y = small_data["y"]
    for e in (orc, hip):
        e.set_state(alpha=a0)
        e.set_residual(y)
        e.sub_xalpha()
    np.testing.assert_array_equal(hip.get_residual(), orc.get_residual())
    np.testing.assert_allclose(hip.mul_alpha(), orc.mul_alpha(), rtol=0, atol=1e-6)
    vare, varg = _hyper(small_data)
    for it in range(1, 6):
        orc.sweep(iteration=it, seed=1, vare=vare, var_effect=varg, pi=0.9)
        hip.sweep(iteration=it, seed=1, vare=vare, var_effect=varg, pi=0.9)
        orc.accumulate(it)
        hip.accumulate(it)
    for mo, mh in zip(orc.posterior(), hip.posterior()):
        np.testing.assert_allclose(mh, mo, rtol=0, atol=1e-6)


def test_alpha_sparse_and_set_columns(hip, small_data):
    """jwas_hip_get_alpha_sparse: the nonzero effects compacted on the device (marker order) -- the binary sample record;
    jwas_hip_set_columns: a matrix uploaded in column chunks equals the one uploaded at once."""
    X = small_data["X"]
    n, p = X.shape
    hip.load_dense(X); hip.setup_blocks(64, "f64"); hip.init_state("BayesC")
    rng = np.random.default_rng(5)
    for frac in (0.0, 0.02, 0.6, 1.0):
        a = np.where(rng.random(p) < frac, rng.standard_normal(p), 0.0).astype(np.float32)
        hip.set_state(alpha=a)
        idx, val = hip.alpha_sparse()
        assert np.array_equal(idx, np.flatnonzero(a)) and np.array_equal(val, a[a != 0])
        np.testing.assert_allclose(hip.mul_alpha(), X.astype(np.float64) @ a.astype(np.float64), atol=2e-4)
    xpx = hip.xpx()
    hip.alloc_dense(n, p)
    for j0 in range(0, p, 100):
        hip.set_columns(j0, X[:, j0:j0 + 100])
    assert np.array_equal(hip.get_columns(0, p), X)
    hip.setup_blocks(64, "f64")
    assert np.array_equal(hip.xpx(), xpx)


def test_state_machine_errors(hip, small_data):
    import jwas_jl_amd as J
    e = J.HipEngine(0)
    with pytest.raises(J.JwasHipError):
        e.setup_blocks(64)
    e.load_dense(small_data["X"])
    with pytest.raises(J.JwasHipError, match="block_size"):
        e.setup_blocks(100)
    e.setup_blocks(64, "f64")
    with pytest.raises(J.JwasHipError, match="init_state"):
        e._chk(e._L.jwas_hip_accumulate(e._h, 1.0))
    with pytest.raises(J.JwasHipError):
        e.init_state("MTBayesC", 9)
    e.close()


@pytest.mark.parametrize("sampler,bs,nreps", [("MTBayesC", 64, 1), ("MTBayesC", 512, 1), ("MTBayesC_II", 128, 1), ("MTBayesC_II", 256, 2), ("MTBayesC", 128, 3)])
def test_mt_marker_specific_prior_parity(hip, sampler, bs, nreps):
    """Marker-specific joint-state priors (the reference's annotated 2-trait BayesC: MarkerSpecificPiPrior,
    MTBayesABC.jl:22-47; p x 4 log prior matrix) against the oracle."""
    t = 2
    data = make_dataset(n=400, p=2 * bs + 29, ncausal=8, seed=77)
    orc, hip = _pair(hip, data, bs, sampler, ntraits=t)
    rng = np.random.default_rng(5)
    Y = np.stack([data["y"] - data["y"].mean() + 0.3 * rng.standard_normal(len(data["y"])).astype(np.float32)
                  for _ in range(t)]).astype(np.float32)
    for k in range(t):
        orc.set_residual(Y[k], k)
        hip.set_residual(Y[k], k)
        if sampler == "MTBayesC":
            ones = np.ones(orc.p, dtype=np.float32)
            orc.set_state(k, delta=ones)
            hip.set_state(k, delta=ones)
    vare = np.array([[0.6, 0.1], [0.1, 0.5]], dtype=np.float32)
    varg = np.array([[0.003, 0.001], [0.001, 0.002]], dtype=np.float32)
    prior = rng.dirichlet(np.array([8.0, 1.0, 1.0, 1.0]), size=orc.p)          # every marker its own prior
    prior[::7] = [0.25, 0.25, 0.25, 0.25]
    lp = np.log(prior)
    for it in range(1, 11):
        so = orc.sweep(iteration=it, seed=21, vare=vare, var_effect=varg, log_prior_states=lp, nreps=nreps)
        sh = hip.sweep(iteration=it, seed=21, vare=vare, var_effect=varg, log_prior_states=lp, nreps=nreps)
        assert np.array_equal(so["state_counts"], sh["state_counts"]), f"iteration {it}"
        np.testing.assert_allclose(sh["beta_ss"], so["beta_ss"], rtol=1e-5)
    assert so["state_counts"][1:].sum() > 0
    for k in range(t):
        _compare_state(orc, hip, k, atol=5e-6)


def test_mt_marker_specific_prior_error_contracts(hip):
    import jwas_jl_amd as J
    data = make_dataset(n=100, p=1100, ncausal=3, seed=1)
    hip.load_dense(data["X"]); hip.setup_blocks(1024, "f64"); hip.init_state("MTBayesC", 2)
    lp = np.log(np.full((1100, 4), 0.25))
    with pytest.raises(J.JwasHipError, match="block size <= 512"):
        hip.sweep(iteration=1, seed=1, vare=np.eye(2, dtype=np.float32), var_effect=np.eye(2, dtype=np.float32) * 0.01, log_prior_states=lp)
    hip.setup_blocks(128, "f64"); hip.init_state("MTBayesC", 3)
    with pytest.raises(J.JwasHipError, match="support 2 traits"):
        hip.sweep(iteration=1, seed=1, vare=np.eye(3, dtype=np.float32), var_effect=np.eye(3, dtype=np.float32) * 0.01,
                  log_prior_states=np.log(np.full((1100, 8), 0.125)))


@pytest.mark.parametrize("spg", [7, 5, 4])
def test_row_group_geometry_does_not_change_the_chain(spg, monkeypatch):
    """Tall matrices run with fewer than 8 streaming waves per row group (224 workgroups of 7 waves at n = 50 000, 5 at
    n = 280 000; jwas_hip.hip alloc_storage).  The geometry only regroups the fp64 partial sums: forced here on a
    small matrix (JWAS_HIP_SPG is read when the storage is allocated), the chain must still equal the oracle's."""
    import jwas_jl_amd as J
    monkeypatch.setenv("JWAS_HIP_SPG", str(spg))
    data = make_dataset(n=2600, p=3 * 256 + 17, ncausal=9, seed=spg)        # 11 slices: ragged last row group
    orc = OracleEngine("lookahead")
    hip = J.HipEngine(0)
    try:
        for e in (orc, hip):
            e.load_dense(data["X"]); e.setup_blocks(256, "f64"); e.init_state("BayesC")
            e.set_residual(data["y"] - data["y"].mean())
        vare, varg = _hyper(data)
        for it in range(1, 9):
            so = orc.sweep(iteration=it, seed=3, vare=vare, var_effect=varg, pi=0.9)
            sh = hip.sweep(iteration=it, seed=3, vare=vare, var_effect=varg, pi=0.9)
            assert so["sum_delta"][0] == sh["sum_delta"][0], f"iteration {it}"
        _compare_state(orc, hip, atol=5e-6)
        for it in range(9, 12):                                              # independent-block kernels on the same geometry
            so = orc.sweep(iteration=it, seed=3, vare=vare, var_effect=varg, pi=0.9, independent_blocks=True)
            sh = hip.sweep(iteration=it, seed=3, vare=vare, var_effect=varg, pi=0.9, independent_blocks=True)
            assert so["sum_delta"][0] == sh["sum_delta"][0], f"independent, iteration {it}"
        _compare_state(orc, hip, atol=5e-6)
        # multi-trait on the same geometry
        for e in (orc, hip):
            e.init_state("MTBayesC", 2)
            for k in range(2):
                e.set_residual((data["y"] - data["y"].mean()) * (1.0 + 0.3 * k), k)
                e.set_state(k, delta=np.ones(e.p, dtype=np.float32))
        lp = np.log(np.array([0.7, 0.1, 0.1, 0.1]))
        V = np.array([[0.6, 0.1], [0.1, 0.5]], dtype=np.float32)
        Gm = np.array([[0.003, 0.001], [0.001, 0.002]], dtype=np.float32)
        for it in range(1, 6):
            so = orc.sweep(iteration=it, seed=4, vare=V, var_effect=Gm, log_prior_states=lp)
            sh = hip.sweep(iteration=it, seed=4, vare=V, var_effect=Gm, log_prior_states=lp)
            assert np.array_equal(so["state_counts"], sh["state_counts"]), f"iteration {it}"
        for k in range(2):
            _compare_state(orc, hip, k, atol=5e-6)
    finally:
        hip.close()


@pytest.mark.parametrize("method,t,bs,sparse", [("MTBayesC", 3, 128, False), ("MTBayesC", 2, 64, True), ("MegaBayesC", 3, 128, False),
                                                ("MTBayesC", 3, 256, False), ("MTBayesC", 2, 256, True),      # (256-marker blocks: dense_big_mt)
                                                ("MTBayesB", 3, 128, False), ("MegaBayesB", 2, 64, True)])
def test_dense_walk_only_multitrait_instantiation_is_bit_identical(method, t, bs, sparse, monkeypatch):
    """The multi-trait sampler's dense-walk-only instantiation (sampler_role_mt<.., DW>: every block walked marker by marker,
    candidacy evaluation / prefix skip / row staging / speculative rounds compiled out; the host picks it after a sweep in which
    most markers changed) against the general instantiation, forced on and off (JWAS_HIP_DENSE_MT): the same chain BIT FOR
    BIT -- also from a sparse state, where the general instantiation runs speculative rounds and the walk evaluates the markers
    outside the model the general way -- with a ragged last block, and against the oracle."""
    import jwas_jl_amd as J
    data = make_dataset(n=700, p=3 * bs + 37, ncausal=9, seed=31 + t)
    y = data["y"] - data["y"].mean()
    rng = np.random.default_rng(3)
    p = data["X"].shape[1]
    A = rng.standard_normal((t, t))
    vare = ((A @ A.T / t + np.eye(t)) * 0.5).astype(np.float32)
    mega = method.startswith("Mega")
    if mega:
        vare = np.diag(np.diag(vare))
        kw = dict(vare=vare, var_effect=(np.eye(t) * 0.003).astype(np.float32), pi=np.full(t, 0.9 if sparse else 0.02))
    else:
        prior = np.full(1 << t, 0.1 / ((1 << t) - 1)) if sparse else np.full(1 << t, 1e-3)
        prior[0 if sparse else -1] = 0.9 if sparse else 1.0
        prior /= prior.sum()
        kw = dict(vare=vare, var_effect=(np.eye(t) * 0.003).astype(np.float32), log_prior_states=np.log(prior))
    if method in ("MTBayesB", "MegaBayesB"):
        Vm = np.zeros((p, t, t), dtype=np.float32)
        for k in range(t):
            Vm[:, k, k] = 0.003 * np.exp(rng.uniform(-1, 1, p))
        kw["var_effect_matrix"] = Vm
    results = {}
    for tag, env in (("oracle", None), ("walk", "1"), ("general", "0")):
        if env is not None:
            monkeypatch.setenv("JWAS_HIP_DENSE_MT", env)
        e = OracleEngine("lookahead") if env is None else J.HipEngine(0)
        try:
            e.load_dense(data["X"]); e.setup_blocks(bs, "f64"); e.init_state(method, t)
            for k in range(t):
                e.set_residual(((1 + 0.25 * k) * y).astype(np.float32), k)
                e.set_state(k, delta=np.ones(p, dtype=np.float32))
            ev = [e.sweep(iteration=it, seed=19, **kw)["n_events"] for it in range(1, 8)]
            results[tag] = ([e.get_state(k) for k in range(t)], [e.get_residual(k) for k in range(t)], ev)
        finally:
            if env is not None:
                e.close()
    assert results["walk"][2] == results["general"][2] == results["oracle"][2]
    for k in range(t):
        for q in range(3):
            assert np.array_equal(results["walk"][0][k][q], results["general"][0][k][q])
        assert np.array_equal(results["walk"][1][k], results["general"][1][k])
        np.testing.assert_allclose(results["walk"][0][k][0], results["oracle"][0][k][0], atol=5e-6)
        assert np.array_equal(results["walk"][0][k][2], results["oracle"][0][k][2])


@pytest.mark.parametrize("t,mixed", [(3, False), (2, True), (4, False)])
def test_dense_big_multitrait_blocks_equal_the_general_path(t, mixed, monkeypatch):
    """Round 4: full 256-marker blocks of sampler I under a dense prior take dense_big_mt (diagonal Gram tiles in LDS, one wave per
    64-marker section, the rest applied in parallel) -- against the same chain through the general path (JWAS_HIP_DENSE_BIG_OFF)
    BIT FOR BIT, and against the oracle.  mixed: a third of the markers start outside the model for a trait and the prior gives the
    other states real mass, so sections are walked with markers evaluated the general way and speculation misses happen."""
    import jwas_jl_amd as J
    data = make_dataset(n=900, p=4 * 256, ncausal=12, seed=60 + t)
    y = data["y"] - data["y"].mean()
    rng = np.random.default_rng(8)
    p = data["X"].shape[1]
    A = rng.standard_normal((t, t))
    vare = ((A @ A.T / t + np.eye(t)) * 0.5).astype(np.float32)
    prior = np.full(1 << t, 0.02 if mixed else 1e-9); prior[-1] = 1.0; prior /= prior.sum()
    kw = dict(vare=vare, var_effect=(np.eye(t) * 0.003).astype(np.float32), log_prior_states=np.log(prior))
    d0 = np.ones((t, p), dtype=np.float32)
    if mixed:
        d0[rng.integers(0, t, p // 3), rng.choice(p, p // 3, replace=False)] = 0.0
    results = {}
    for tag in ("oracle", "big", "general"):
        if tag == "general":
            monkeypatch.setenv("JWAS_HIP_DENSE_BIG_OFF", "1")
        e = OracleEngine("lookahead") if tag == "oracle" else J.HipEngine(0)
        try:
            e.load_dense(data["X"]); e.setup_blocks(256, "f64"); e.init_state("MTBayesC", t)
            for k in range(t):
                e.set_residual(((1 + 0.25 * k) * y).astype(np.float32), k)
                e.set_state(k, delta=d0[k])
            ev = [e.sweep(iteration=it, seed=23, **kw)["n_events"] for it in range(1, 10)]
            results[tag] = ([e.get_state(k) for k in range(t)], [e.get_residual(k) for k in range(t)], ev)
        finally:
            if tag != "oracle":
                e.close()
    assert results["big"][2] == results["general"][2] == results["oracle"][2]
    for k in range(t):
        for q in range(3):
            assert np.array_equal(results["big"][0][k][q], results["general"][0][k][q])
        assert np.array_equal(results["big"][1][k], results["general"][1][k])
        np.testing.assert_allclose(results["big"][0][k][0], results["oracle"][0][k][0], atol=5e-6)
        assert np.array_equal(results["big"][0][k][2], results["oracle"][0][k][2])


@pytest.mark.parametrize("method,t,bs,pi", [("BayesC", 1, 128, 0.0), ("BayesC", 1, 256, 0.5), ("MTBayesC", 3, 128, None), ("BayesR", 1, 64, None)])
def test_cooperative_dense_apply_is_bit_identical(method, t, bs, pi, monkeypatch):
    """Dense sweeps let the column groups of a row group split the rows when a block's changes are applied to the residual
    (update_role, cooperative dense apply: shares through r_out, an arrival counter per row group, bounded wait with a
    fall-back).  Forced on and off (JWAS_HIP_COOP_APPLY) on a matrix with several row groups and a ragged last slice: the
    chain must be the same BIT FOR BIT either way, and equal the oracle's."""
    import jwas_jl_amd as J
    monkeypatch.setenv("JWAS_HIP_SPG", "4")
    data = make_dataset(n=2300, p=4 * bs + 19, ncausal=12, seed=5 + t)       # 9 slices -> 3 row groups of <= 4 waves
    y = data["y"] - data["y"].mean()
    rng = np.random.default_rng(8)
    vare1, varg1 = _hyper(data)
    if t == 1:
        kw = dict(vare=vare1, var_effect=varg1)
        if method == "BayesR":
            kw["pi_classes"] = np.array([0.3, 0.3, 0.2, 0.2])
        else:
            kw["pi"] = pi
    else:
        A = rng.standard_normal((t, t))
        prior = np.full(1 << t, 1e-3); prior[-1] = 1.0; prior /= prior.sum()
        kw = dict(vare=(A @ A.T / t + np.eye(t)).astype(np.float32) * 0.5, var_effect=(np.eye(t) * 0.002).astype(np.float32),
                  log_prior_states=np.log(prior))
    results = {}
    for tag, env in (("oracle", None), ("coop", "1"), ("plain", "0")):
        if env is not None:
            monkeypatch.setenv("JWAS_HIP_COOP_APPLY", env)
        e = OracleEngine("lookahead") if env is None else J.HipEngine(0)
        try:
            e.load_dense(data["X"]); e.setup_blocks(bs, "f64"); e.init_state(method, t)
            for k in range(t):
                e.set_residual(((1 + 0.25 * k) * y).astype(np.float32), k)
            ev = []
            for it in range(1, 7):
                ev.append(e.sweep(iteration=it, seed=77, **kw)["n_events"])
            results[tag] = ([e.get_state(k) for k in range(t)], [e.get_residual(k) for k in range(t)], ev)
        finally:
            if env is not None:
                e.close()
    assert results["coop"][2] == results["plain"][2] == results["oracle"][2]
    assert min(results["coop"][2]) >= 32 * 4                                 # dense enough for the cooperative path in every block
    for k in range(t):
        for q in range(3):
            assert np.array_equal(results["coop"][0][k][q], results["plain"][0][k][q])
        assert np.array_equal(results["coop"][1][k], results["plain"][1][k])
        np.testing.assert_allclose(results["coop"][1][k], results["oracle"][1][k], atol=2e-4)
        np.testing.assert_allclose(results["coop"][0][k][0], results["oracle"][0][k][0], atol=5e-6)


@pytest.mark.parametrize("method", ["MTBayesB", "MTBayesB_II"])
@pytest.mark.parametrize("t,bs,dense,nreps", [(2, 64, False, 1), (3, 128, False, 1), (3, 512, False, 1), (4, 256, False, 1),
                                              (3, 128, True, 1), (2, 64, True, 1), (3, 64, False, 3),
                                              (3, 256, True, 1), (2, 256, True, 1)])      # (dense 256-marker blocks: dense_big_mt with the markers' own constants)
def test_mt_bayesb_per_marker_covariance_parity(hip, t, bs, dense, nreps, method):
    """Multi-trait BayesA/B: Gibbs sampler I (MTBayesABC.jl:66,86-90) or the joint-state sampler II (MTBayesABC.jl:129-210)
    with ONE t x t effect covariance PER MARKER (Ginv = inv.(varEffects)).  The device inverts every marker's matrix in k_prepare (the host's Gauss-Jordan, operation for
    operation) and the sampler uses the marker's own constants; sparse rounds, the dense walk (every marker in the model,
    speculative and general) and within-block repetitions against the oracle."""
    import jwas_jl_amd as J
    data = make_dataset(n=420, p=2 * bs + 29, ncausal=10, seed=900 + t)
    orc, hip = _pair(hip, data, bs, method, ntraits=t)
    rng = np.random.default_rng(40 + t)
    p = orc.p
    Y = np.stack([data["y"] - data["y"].mean() + 0.3 * rng.standard_normal(len(data["y"])).astype(np.float32)
                  for _ in range(t)]).astype(np.float32)
    for k in range(t):
        orc.set_residual(Y[k], k); hip.set_residual(Y[k], k)
        ones = np.ones(p, dtype=np.float32)
        if dense:
            a0 = (0.01 * rng.standard_normal(p)).astype(np.float32)
            orc.set_state(k, alpha=a0, beta=a0, delta=ones); hip.set_state(k, alpha=a0, beta=a0, delta=ones)
            orc.sub_xalpha(k); hip.sub_xalpha(k)
        else:
            orc.set_state(k, delta=ones); hip.set_state(k, delta=ones)
    A = rng.standard_normal((t, t))
    vare = (A @ A.T / t + np.eye(t)).astype(np.float32) * 0.5
    # a different SPD matrix for every marker (scales over two orders of magnitude)
    Bm = rng.standard_normal((p, t, t))
    Gm = ((Bm @ Bm.transpose(0, 2, 1) / t + np.eye(t)) * (0.002 * np.exp(rng.uniform(-2, 2, p)))[:, None, None]).astype(np.float32)
    if dense:
        prior = np.full(1 << t, 1e-4); prior[-1] = 1.0; prior /= prior.sum()
    else:
        prior = rng.dirichlet(np.ones(1 << t))
    kw = dict(vare=vare, var_effect=np.eye(t, dtype=np.float32), var_effect_matrix=Gm, log_prior_states=np.log(prior), nreps=nreps)
    for it in range(1, 13):
        so = orc.sweep(iteration=it, seed=13, **kw)
        sh = hip.sweep(iteration=it, seed=13, **kw)
        assert np.array_equal(so["state_counts"], sh["state_counts"]), f"iteration {it}"
        np.testing.assert_allclose(sh["beta_ss"], so["beta_ss"], rtol=1e-5)
        np.testing.assert_allclose(sh["resid_ss"], so["resid_ss"], rtol=1e-5, atol=1e-6 * float(np.max(np.abs(so["resid_ss"]))))
    for k in range(t):
        _compare_state(orc, hip, k, atol=5e-6)
    # error contracts
    with pytest.raises(J.JwasHipError, match="independent_blocks"):
        hip.sweep(iteration=1, seed=1, independent_blocks=True, **kw)
    # var_effect_matrix=None means "the covariances resident on the device" (an earlier sweep's, or those drawn by
    # jwas_hip_sample_marker_covariances): same chain as handing the same matrices over again ...
    a_before = [hip.get_state(k)[0].copy() for k in range(t)]
    kw_res = dict(kw); kw_res["var_effect_matrix"] = None
    orc.sweep(iteration=13, seed=13, **kw); hip.sweep(iteration=13, seed=13, **kw_res)
    for k in range(t):
        _compare_state(orc, hip, k, atol=5e-6)
    assert any(not np.array_equal(a_before[k], hip.get_state(k)[0]) for k in range(t))
    # ... and an error while none are resident (a fresh state)
    hip.init_state(method, t)
    with pytest.raises(J.JwasHipError, match="per-marker effect covariances"):
        hip.sweep(iteration=1, seed=1, **kw_res)


@pytest.mark.parametrize("t,bs,pi,nreps", [(2, 64, 0.9, 1), (3, 128, 0.5, 1), (4, 256, 0.95, 1), (3, 128, 0.0, 1), (2, 64, 0.7, 3)])
def test_mega_bayesb_per_marker_variances_parity(hip, t, bs, pi, nreps):
    """megaBayesABC! with BayesA/B (BayesABC.jl:1-8: [vari[i,i] for vari in locus_effect_variances]; G.constraint = true): t
    independent single-trait chains, trait k of marker j under its own variance.  Sparse and dense (pi = 0: BayesA) against
    the oracle; the off-diagonal entries of the matrices handed over are ignored; and the constrained draw of the next
    iteration's variances, G_kk = (scale_kk + b_jk^2) / chi2(df) (variance_components.jl:112-117), on the device."""
    import oracle as O
    data = make_dataset(n=380, p=2 * bs + 17, ncausal=8, seed=700 + t)
    orc, hip = _pair(hip, data, bs, "MegaBayesB", ntraits=t)
    rng = np.random.default_rng(70 + t)
    p = orc.p
    for k in range(t):
        y = (data["y"] - data["y"].mean() + 0.3 * rng.standard_normal(len(data["y"]))).astype(np.float32)
        orc.set_residual(y, k); hip.set_residual(y, k)
        ones = np.ones(p, dtype=np.float32)
        orc.set_state(k, delta=ones); hip.set_state(k, delta=ones)
    vare = np.diag(rng.uniform(0.3, 0.8, t)).astype(np.float32)
    Vm = np.zeros((p, t, t), dtype=np.float32)
    for k in range(t):
        Vm[:, k, k] = 0.003 * np.exp(rng.uniform(-2, 2, p))
    Vh = Vm.copy()
    Vh[:, 0, t - 1] = 7.0; Vh[:, t - 1, 0] = -3.0                       # (ignored: only the diagonal is read)
    kw = dict(vare=vare, var_effect=np.eye(t, dtype=np.float32), pi=np.full(t, pi), nreps=nreps)
    for it in range(1, 9):
        so = orc.sweep(iteration=it, seed=5, var_effect_matrix=Vm, **kw)
        sh = hip.sweep(iteration=it, seed=5, var_effect_matrix=Vh if it % 2 else Vm, **kw)
        assert np.array_equal(so["sum_delta"], sh["sum_delta"]), f"iteration {it}"
        np.testing.assert_allclose(sh["resid_ss"], so["resid_ss"], rtol=1e-5, atol=1e-6 * float(np.max(np.abs(so["resid_ss"]))))
    for k in range(t):
        _compare_state(orc, hip, k, atol=5e-6)
    # the constrained draw: diagonal only, same counters as the oracle's
    scale = np.diag(rng.uniform(0.01, 0.05, t))
    hip.sample_marker_covariances(5.5, scale, seed=9, iteration=9, marker_offset=11)
    Gh = hip.marker_covariances()
    beta = np.stack([hip.get_state(k)[1] for k in range(t)])
    Go = O.sample_marker_covariances(beta, 5.5, scale, 9, 9, 11, diagonal=True)
    np.testing.assert_allclose(Gh, Go, rtol=2e-6, atol=0)
    off = ~np.eye(t, dtype=bool)
    assert (Gh[:, off] == 0).all() and (Gh[:, ~off] > 0).all()
    orc._var_mat = Gh                                                    # (the device's own draws: libm's last place aside)
    so = orc.sweep(iteration=9, seed=5, **kw); sh = hip.sweep(iteration=9, seed=5, **kw)      # the resident draws
    assert np.array_equal(so["sum_delta"], sh["sum_delta"])


def test_mt_bayesb_needs_parked_draws(hip):
    import jwas_jl_amd as J
    data = make_dataset(n=100, p=1100, ncausal=3, seed=1)
    hip.load_dense(data["X"]); hip.setup_blocks(1024, "f64"); hip.init_state("MTBayesB", 3)
    Gm = np.tile(np.eye(3, dtype=np.float32) * 0.01, (1100, 1, 1))
    with pytest.raises(J.JwasHipError, match="block_size \\* ntraits <= 2048"):
        hip.sweep(iteration=1, seed=1, vare=np.eye(3, dtype=np.float32), var_effect=np.eye(3, dtype=np.float32),
                  var_effect_matrix=Gm, log_prior_states=np.log(np.full(8, 0.125)))


@pytest.mark.parametrize("method,t,nreps,gram", [("BayesC", 1, 0, "f64"), ("BayesC", 1, 1, "mfma"), ("BayesR", 1, 0, "f64"),
                                                 ("MTBayesC", 2, 0, "f64"), ("MTBayesC_II", 2, 1, "f64"), ("BayesC", 1, 3, "f64")])
def test_explicit_non_uniform_block_starts_parity(hip, method, t, nreps, gram):
    """fast_blocks = a vector of block starts (JWAS.jl:298-304): blocks of different sizes (here 1 ... 200 markers, some
    larger than one 64-marker sub-block, a single-marker block, a short last block), every block running its own size as
    repetition count when nreps <= 0 (BayesABC.jl:153).  Grams / cross-Grams of the ragged partition and the whole chain
    against the oracle on the same partition."""
    import jwas_jl_amd as J
    data = make_dataset(n=360, p=700, ncausal=10, seed=61)
    starts = np.array([0, 70, 71, 200, 400, 430, 560, 690], dtype=np.int64)
    orc = OracleEngine("lookahead")
    orc.load_dense(data["X"]); orc.setup_blocks_explicit(starts); orc.init_state(method, t)
    hip.load_dense(data["X"]); hip.setup_blocks_explicit(starts, gram); hip.init_state(method, t)
    assert hip.block_size == 256 and hip.nblocks == len(starts)
    sizes = np.diff(np.append(starts, 700))
    for k in (0, 1, 3, 7):
        g = hip.gram(k)
        assert g.shape == (sizes[k], sizes[k])
        np.testing.assert_allclose(g, orc._grams[int((sizes[:k] ** 2).sum()):int((sizes[:k + 1] ** 2).sum())].reshape(sizes[k], sizes[k]),
                                   rtol=(1e-6 if gram == "f64" else 2e-5), atol=(1e-6 if gram == "f64" else 2e-3))
    y = data["y"] - data["y"].mean()
    rng = np.random.default_rng(6)
    for k in range(t):
        yk = ((1 + 0.3 * k) * y + 0.1 * k * rng.standard_normal(len(y))).astype(np.float32)
        orc.set_residual(yk, k); hip.set_residual(yk, k)
        if t > 1:
            ones = np.ones(700, dtype=np.float32)
            orc.set_state(k, delta=ones); hip.set_state(k, delta=ones)
    vare1, varg1 = _hyper(data)
    if method == "BayesR":
        kw = dict(vare=vare1, var_effect=np.float32(varg1 * 5), pi_classes=np.array([0.9, 0.05, 0.03, 0.02]))
    elif t == 1:
        kw = dict(vare=vare1, var_effect=varg1, pi=0.9)
    else:
        kw = dict(vare=(np.eye(t) * 0.5 + 0.1).astype(np.float32), var_effect=(np.eye(t) * 0.004).astype(np.float32),
                  log_prior_states=np.log(np.array([0.7, 0.1, 0.1, 0.1])))
    for it in range(1, 9):
        so = orc.sweep(iteration=it, seed=9, nreps=nreps, **kw)
        sh = hip.sweep(iteration=it, seed=9, nreps=nreps, **kw)
        assert so["n_events"] == sh["n_events"], f"iteration {it}"
    tol = 5e-6 if gram == "f64" else 5e-4
    for k in range(t):
        if gram == "f64":
            _compare_state(orc, hip, k, atol=tol)
        else:
            np.testing.assert_allclose(hip.get_state(k)[0], orc.get_state(k)[0], atol=tol)
    # independent blocks on the same ragged partition (BayesABC_block_independent!, BayesABC.jl:190-255, under explicit starts):
    # every block's rhs from the residual snapshot, blocks of 1 ... 200 markers sampled concurrently
    for it in range(9, 14):
        so = orc.sweep(iteration=it, seed=9, nreps=nreps, independent_blocks=True, **kw)
        sh = hip.sweep(iteration=it, seed=9, nreps=nreps, independent_blocks=True, **kw)
        assert so["n_events"] == sh["n_events"], f"independent iteration {it}"
    for k in range(t):
        if gram == "f64":
            _compare_state(orc, hip, k, atol=tol)
            np.testing.assert_allclose(hip.get_residual(k), orc.get_residual(k), atol=2e-4)
        else:
            np.testing.assert_allclose(hip.get_state(k)[0], orc.get_state(k)[0], atol=tol)
    with pytest.raises(J.JwasHipError, match="explicit block partition"):
        hip.add_block_size(512, "f64")
    with pytest.raises(J.JwasHipError, match="sorted, unique"):
        hip.setup_blocks_explicit(np.array([0, 50, 50, 100]), "f64")
    with pytest.raises(J.JwasHipError, match="at most 1024 markers"):
        hip.load_dense(np.zeros((8, 1300), dtype=np.float32)); hip.setup_blocks_explicit(np.array([0, 1100]), "f64")


@pytest.mark.parametrize("method", ["BayesC", "BayesR"])
def test_block_repetitions_with_blocks_that_start_without_candidates(hip, method):
    """Within-block repetitions on a sparse chain: most 64-marker blocks hold no candidate at entry (the single-pass sampler
    skips such blocks entirely), but a later repetition draws anew and may move a marker -- whose Gram row must then be
    fetched, not looked up through a slot left over from an earlier block.  Many blocks, many repetitions, several block
    sizes alternating in one process (stale LDS from the previous configuration is the hazard)."""
    data = make_dataset(n=300, p=64 * 14 + 9, ncausal=4, seed=71)
    vare, varg = _hyper(data)
    for bs, nreps in ((256, 1), (64, 12), (128, 0), (64, 24)):
        orc, hip = _pair(hip, data, bs, method)
        r0 = data["y"] - data["y"].mean()
        orc.set_residual(r0); hip.set_residual(r0)
        if method == "BayesR":
            kw = dict(var_effect=np.float32(20 * varg), pi_classes=np.array([0.992, 0.004, 0.003, 0.001]))
            for e in (orc, hip):
                e.set_state(delta=np.ones(e.p, dtype=np.int32))
        else:
            kw = dict(var_effect=np.float32(8 * varg), pi=0.992)
        for it in range(1, 5):
            so = orc.sweep(iteration=it, seed=17, vare=vare, nreps=nreps, **kw)
            sh = hip.sweep(iteration=it, seed=17, vare=vare, nreps=nreps, **kw)
            assert so["n_events"] == sh["n_events"], f"bs={bs} nreps={nreps} iteration {it}"
        ao, _, do = orc.get_state()
        ah, _, dh = hip.get_state()
        assert np.array_equal(do, dh)
        np.testing.assert_allclose(ah, ao, rtol=0, atol=5e-6)


@pytest.mark.parametrize("method,t,bs,nreps,n,weights", [("BayesC", 1, 64, 1, 700, False), ("BayesC", 1, 512, 0, 664, False), ("BayesR", 1, 256, 1, 1500, True),
                                                         ("MTBayesC", 3, 128, 1, 900, False), ("BayesC", 1, 256, 300, 2600, False),
                                                         ("MTBayesC_II", 2, 64, 2, 500, False)])
def test_every_bit_equal_when_the_oracle_sums_in_the_device_order(hip, method, t, bs, nreps, n, weights):
    """The device and the oracle perform the same operations on the same numbers with ONE exception: the association of the
    fp64 additions in a block's x'r (row groups on the device, row by row in the oracle).  With the oracle summing in the
    device's order (ORC_ACC_DEVICE, the geometry from jwas_hip_update_geometry) and the same precomputed inner products
    on both sides (x'x, Grams, cross-Grams handed to the device), every effect, every indicator and every residual element
    must be equal bit for bit -- also after hundreds of within-block repetitions, where last-bit differences would be
    amplified (the case the fuzz run found: 512 repetitions, n = 664, p = 704)."""
    import oracle as O
    p = 704 if bs == 512 else 3 * bs + 37
    d = make_dataset(n=n, p=p, ncausal=8, seed=2319 % 1000 if bs == 512 else 5)
    y = (d["y"] - d["y"].mean()).astype(np.float32)
    rng = np.random.default_rng(3)
    w = rng.uniform(0.4, 2.5, n).astype(np.float32) if weights else None
    hip.load_dense(d["X"]); hip.set_weights(w); hip.setup_blocks(bs, "f64"); hip.init_state(method, t)
    spg, nrg, ncg = hip.update_geometry()
    O.set_device_order(spg)
    orc = OracleEngine("lookahead", acc=O.ACC_DEVICE)
    orc.load_dense(d["X"]); orc.set_weights(w); orc.setup_blocks(bs); orc.init_state(method, t)
    # the same precomputed inner products on both sides
    hip.set_xpx(orc._xpx)
    hip.set_grams_packed(orc._grams)
    orc._w()
    starts = list(orc._bs) + [p]
    for k in range(1, len(starts) - 1):
        hip.set_cross_gram(k, O.cross_gram(d["X"], starts[k - 1], starts[k] - starts[k - 1], starts[k], starts[k + 1] - starts[k], O.ACC_DEVICE))
    O.set_weights(None)
    for k in range(t):
        yk = ((1 + 0.3 * k) * y).astype(np.float32)
        orc.set_residual(yk, k); hip.set_residual(yk, k)
        if method == "BayesR":
            orc.set_state(0, delta=np.ones(p, dtype=np.int32)); hip.set_state(0, delta=np.ones(p, dtype=np.int32))
        elif t > 1:
            orc.set_state(k, delta=np.ones(p, dtype=np.float32)); hip.set_state(k, delta=np.ones(p, dtype=np.float32))
    v = np.float32(max(float(np.var(y)), 0.1))
    if method == "BayesR":
        kw = dict(vare=v, var_effect=np.float32(0.1), pi_classes=np.array([0.6, 0.2, 0.1, 0.1]))
    elif t == 1:
        kw = dict(vare=v, var_effect=np.float32(0.02), pi=0.3)
    else:
        A = rng.standard_normal((t, t)); B = rng.standard_normal((t, t))
        kw = dict(vare=((A @ A.T / t + np.eye(t)) * v).astype(np.float32), var_effect=((B @ B.T / t + np.eye(t)) * 0.02).astype(np.float32),
                  log_prior_states=np.log(rng.dirichlet(np.ones(1 << t))))
    try:
        for it in range(1, 7):
            so = orc.sweep(iteration=it, seed=2319, nreps=nreps, **kw)
            sh = hip.sweep(iteration=it, seed=2319, nreps=nreps, **kw)
            assert so["n_events"] == sh["n_events"], f"iteration {it}"
            for k in range(t):
                for q, (xo, xh) in enumerate(zip(orc.get_state(k), hip.get_state(k))):
                    assert np.array_equal(xo, xh), f"iteration {it}, trait {k}, field {q}: max diff {np.abs(xo.astype(np.float64) - xh).max()}"
                assert np.array_equal(orc.get_residual(k), hip.get_residual(k)), f"iteration {it}: residual of trait {k}"
    finally:
        O.set_device_order(8)
        hip.set_weights(None)


def test_cooperative_dense_apply_stress_at_production_geometry(monkeypatch):
    """Stress run of the cooperative dense apply (sweep.hpp update_role<COOP>): the protocol publishes a row group's shares
    with relaxed agent-scope stores + s_waitcnt, counts arrivals in a relaxed counter and reads the peers' shares with
    relaxed agent-scope loads -- it relies on the hardware's ordering of a wave's own stores, so a stale read would be a
    silently wrong residual.  >= 10 000 cooperative launches at the production geometry (n = 50 000: 28 row groups x 8
    column groups, every CU busy, Pi = 0 so every marker of every block changes), residual and effects compared BIT FOR
    BIT with the redundant apply (JWAS_HIP_COOP_APPLY=0) every ten sweeps."""
    import os
    import jwas_jl_amd as J
    n, p, bs, sweeps = 50_000, 12_800, 128, 100
    engs = {}
    try:
        for tag in ("coop", "plain"):
            e = J.HipEngine(0)
            e.alloc_dense(n, p)
            e.synth(11, kind=0, center=True)
            e.setup_blocks(bs, "mfma")
            e.init_state("BayesC", 1)
            engs[tag] = e
        rng = np.random.default_rng(3)
        y = rng.standard_normal(n).astype(np.float32)
        for e in engs.values():
            e.set_residual(y)
        launches = 0
        for it in range(1, sweeps + 1):
            for tag, e in engs.items():
                monkeypatch.setenv("JWAS_HIP_COOP_APPLY", "1" if tag == "coop" else "0")      # (read once per sweep)
                st = e.sweep(iteration=it, seed=5, vare=np.float32(1.0), var_effect=np.float32(1e-4), pi=0.0)
                assert st["n_events"] == p
            launches += p // bs + 1
            if it % 10 == 0:
                assert np.array_equal(engs["coop"].get_residual(), engs["plain"].get_residual()), f"residuals differ after sweep {it}"
                assert np.array_equal(engs["coop"].get_state()[0], engs["plain"].get_state()[0]), f"effects differ after sweep {it}"
        assert launches >= 10_000
        # and the residual still is y - X alpha (what a stale share would break first)
        e = engs["coop"]
        r = e.get_residual().astype(np.float64)
        xa = e.mul_alpha(0).astype(np.float64)
        assert np.abs((y.astype(np.float64) - xa) - r).max() < 5e-3
    finally:
        for e in engs.values():
            e.close()


@pytest.mark.parametrize("method,bs,n,p", [("BayesC", 256, 900, 256 * 4 + 77), ("BayesC", 512, 2300, 512 * 3 + 130),
                                           ("BayesA", 512, 700, 512 * 2), ("BayesC", 256, 300, 256 * 3)])
def test_dense_big_blocks_parity(method, bs, n, p, monkeypatch):
    """Pi = 0 on 256- / 512-marker blocks: the sampler's dense_big_st path (sweep.hpp) -- 64-marker sections walked by one
    wave each from LDS-resident diagonal Gram tiles, everything off the diagonal applied in parallel from prefetched
    registers, the lookahead correction accumulated section by section.  It performs the sequential chain's own fmaf
    sequence, so the chain must equal the oracle's lookahead form (indicators exactly, effects to the rounding of x'r) and
    must equal BIT FOR BIT what the general path gives (JWAS_HIP_DENSE_BIG_OFF=1); the ragged last block runs the general
    path inside the same sweep."""
    import jwas_jl_amd as J
    monkeypatch.setenv("JWAS_HIP_SPG", "4")                                  # several row groups also on short matrices
    data = make_dataset(n=n, p=p, ncausal=10, seed=41)
    y = (data["y"] - data["y"].mean()).astype(np.float32)
    rng = np.random.default_rng(2)
    kw = dict(vare=np.float32(0.7), var_effect=np.float32(0.003), pi=0.0)
    code = "BayesB" if method == "BayesA" else method
    if method == "BayesA":
        kw["var_effect_vec"] = rng.uniform(0.001, 0.01, size=p).astype(np.float32)
    res = {}
    for tag in ("oracle", "big", "general"):
        if tag == "general":
            monkeypatch.setenv("JWAS_HIP_DENSE_BIG_OFF", "1")
        e = OracleEngine("lookahead") if tag == "oracle" else J.HipEngine(0)
        try:
            e.load_dense(data["X"]); e.setup_blocks(bs, "f64"); e.init_state(code, 1)
            e.set_residual(y)
            ev = [e.sweep(iteration=it, seed=21, **kw)["n_events"] for it in range(1, 6)]
            res[tag] = (e.get_state(0), e.get_residual(0), ev)
        finally:
            if tag != "oracle":
                e.close()
    assert res["big"][2] == res["general"][2] == res["oracle"][2] == [float(p)] * 5
    for q in range(3):
        assert np.array_equal(res["big"][0][q], res["general"][0][q])
    assert np.array_equal(res["big"][1], res["general"][1])
    assert np.array_equal(res["big"][0][2], res["oracle"][0][2])
    np.testing.assert_allclose(res["big"][0][0], res["oracle"][0][0], atol=5e-6)
    np.testing.assert_allclose(res["big"][1], res["oracle"][1], atol=2e-4)


def test_dense_big_blocks_every_bit_equal_the_oracle(hip):
    """dense_big_st against the oracle with the oracle summing x'r in the device's order and the device using the oracle's
    inner products: every effect and residual element equal bit for bit (the same standard as
    test_every_bit_equal_when_the_oracle_sums_in_the_device_order), BayesC Pi = 0 on 512-marker blocks, two full blocks."""
    import oracle as O
    n, p, bs = 1100, 512 * 2, 512
    d = make_dataset(n=n, p=p, ncausal=10, seed=43)
    y = (d["y"] - d["y"].mean()).astype(np.float32)
    hip.load_dense(d["X"]); hip.setup_blocks(bs, "f64"); hip.init_state("BayesC", 1)
    spg, nrg, ncg = hip.update_geometry()
    O.set_device_order(spg)
    orc = OracleEngine("lookahead", acc=O.ACC_DEVICE)
    orc.load_dense(d["X"]); orc.setup_blocks(bs); orc.init_state("BayesC", 1)
    hip.set_xpx(orc._xpx)
    hip.set_grams_packed(orc._grams)
    starts = list(orc._bs) + [p]
    for k in range(1, len(starts) - 1):
        hip.set_cross_gram(k, O.cross_gram(d["X"], starts[k - 1], starts[k] - starts[k - 1], starts[k], starts[k + 1] - starts[k], O.ACC_DEVICE))
    orc.set_residual(y); hip.set_residual(y)
    for it in range(1, 6):
        kw = dict(iteration=it, seed=8, vare=np.float32(0.6), var_effect=np.float32(0.002), pi=0.0)
        so = orc.sweep(**kw); sh = hip.sweep(**kw)
        assert so["n_events"] == sh["n_events"] == p
        assert np.array_equal(hip.get_state(0)[0], orc.get_state(0)[0]), f"iteration {it}"
        assert np.array_equal(hip.get_residual(0), orc.get_residual(0)), f"iteration {it}"


@pytest.mark.parametrize("t", [2, 3, 4])
def test_marker_covariance_draws_device_vs_oracle(hip, t):
    """jwas_hip_sample_marker_covariances: one InverseWishart(df, scale + b_j b_j') draw per marker from the resident beta
    (variance_components.jl:181-186; Bartlett's decomposition on the counter RNG, keyed by the global marker index) against
    the oracle's restatement -- same counters, same operations, double arithmetic rounded to float once: equal to the last
    bit except where libm and the device's log / cos / sqrt differ in their last place -- and against the distribution's
    mean E[G] = (scale + b b') / (df - t - 1)."""
    import oracle as O
    n, p = 300, 6000
    d = make_dataset(n=n, p=p, ncausal=5, seed=17)
    hip.load_dense(d["X"]); hip.setup_blocks(64, "f64"); hip.init_state("MTBayesB", t)
    rng = np.random.default_rng(t)
    beta = (0.05 * rng.standard_normal((t, p))).astype(np.float32)
    for k in range(t):
        hip.set_state(k, alpha=beta[k], beta=beta[k], delta=np.ones(p, dtype=np.float32))
    A = rng.standard_normal((t, t))
    scale = (A @ A.T / t + np.eye(t)) * 0.01
    df = t + 5.0
    hip.sample_marker_covariances(df, scale, seed=11, iteration=4, marker_offset=1000)
    Gh = hip.marker_covariances()
    Go = O.sample_marker_covariances(beta, df, scale, 11, 4, 1000)
    assert np.isfinite(Gh).all()
    np.testing.assert_allclose(Gh, Go, rtol=2e-6, atol=0)
    assert (Gh == Go).mean() > 0.98
    assert np.array_equal(Gh, Gh.transpose(0, 2, 1))
    # a different iteration / marker offset gives different draws; the same call the same draws
    hip.sample_marker_covariances(df, scale, seed=11, iteration=4, marker_offset=1000)
    assert np.array_equal(hip.marker_covariances(), Gh)
    hip.sample_marker_covariances(df, scale, seed=11, iteration=5, marker_offset=1000)
    assert not np.array_equal(hip.marker_covariances(), Gh)
    # mean of the draws (b = 0: E[G] = scale / (df - t - 1))
    for k in range(t):
        hip.set_state(k, beta=np.zeros(p, dtype=np.float32))
    hip.sample_marker_covariances(df, scale, seed=3, iteration=1)
    np.testing.assert_allclose(hip.marker_covariances().mean(axis=0), scale / (df - t - 1), rtol=0.15, atol=2e-4)
    # the next sweep uses the resident draws (no matrix handed over)
    y = (d["y"] - d["y"].mean()).astype(np.float32)
    for k in range(t):
        hip.set_residual(y, k)
    prior = np.full(1 << t, 0.5 / ((1 << t) - 1)); prior[0] = 0.5
    st = hip.sweep(iteration=1, seed=3, vare=np.eye(t, dtype=np.float32) * 0.5, var_effect=np.eye(t, dtype=np.float32),
                   log_prior_states=np.log(prior))
    assert st["state_counts"].sum() == p
