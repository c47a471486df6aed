"""Float64 mode on the GPU (runMCMC(double_precision=true), JWAS.jl:349-366): the device's Float64 context (csrc/f64_path.hpp,
through the C ABI's *_f64 entry points) against the Float64 oracle (oracle/jwas_oracle_f64.c: the reference's scalar kernels
with T = Float64 in the literal per-marker dot / update / axpy order).  Stated tolerance: inclusion / class trajectories
identical; effects and residual within 1e-9 relative (both sides compute in double; only the association of the sums
differs: block form + Gram on the device, per-marker dot products in the oracle)."""
import numpy as np
import pandas as pd
import pytest

from conftest import make_dataset
from oracle_engine import OracleEngine64
from jwas_jl_amd import api

pytestmark = pytest.mark.gpu


def _pair(n, p, bs, method, t=1, seed=3):
    import jwas_jl_amd as J
    d = make_dataset(n=n, p=p, ncausal=8, seed=seed)
    X = np.asfortranarray(d["X"].astype(np.float64))
    hip, orc = J.HipEngine(0, precision=64), OracleEngine64()
    for e in (orc, hip):
        e.load_dense(X); e.setup_blocks(bs); e.init_state(method, t)
    return d, X, orc, hip


def _compare(orc, hip, t):
    for k in range(t):
        ao, bo, do = orc.get_state(k)
        ah, bh, dh = hip.get_state(k)
        assert np.array_equal(do, dh), f"trait {k}: indicators differ at {np.flatnonzero(do != dh)[:5]}"
        np.testing.assert_allclose(ah, ao, rtol=1e-9, atol=1e-12)
        np.testing.assert_allclose(bh, bo, rtol=1e-9, atol=1e-12)
        np.testing.assert_allclose(hip.get_residual(k), orc.get_residual(k), rtol=0, atol=1e-9)


def test_f64_xpx_and_dtype_contract():
    import jwas_jl_amd as J
    d, X, orc, hip = _pair(300, 200, 64, "BayesC")
    try:
        assert hip.xpx().dtype == np.float64
        np.testing.assert_allclose(hip.xpx(), (X * X).sum(axis=0), rtol=1e-13)
        with pytest.raises(TypeError, match="float64"):
            hip.load_dense(X.astype(np.float32))
        with pytest.raises(J.JwasHipError, match="1 to 1024 markers"):
            hip.setup_blocks(2048)
        with pytest.raises(J.JwasHipError, match="Float64 context"):
            hip.add_block_size(512)
        f32 = J.HipEngine(0)
        try:
            with pytest.raises(TypeError, match="float32"):
                f32.load_dense(X)
        finally:
            f32.close()
    finally:
        hip.close()


@pytest.mark.parametrize("method,bs,n,p", [("BayesC", 128, 700, 128 * 3 + 50), ("BayesC", 64, 300, 64 * 5 + 7), ("BayesA", 128, 520, 400),
                                           ("BayesB", 128, 257, 300)])
def test_f64_bayesabc_parity(method, bs, n, p):
    d, X, orc, hip = _pair(n, p, bs, "BayesB" if method in ("BayesA", "BayesB") else "BayesC")
    try:
        y = d["y"] - d["y"].mean()
        rng = np.random.default_rng(5)
        kw = dict(vare=0.55, var_effect=0.004)
        if method == "BayesC":
            kw["pi"] = rng.uniform(0.6, 0.99, p) if bs == 64 else 0.9          # per-marker pi on one case
        else:
            kw["pi"] = 0.0 if method == "BayesA" else 0.8
            kw["var_effect_vec"] = rng.uniform(0.001, 0.01, p)
        for e in (orc, hip):
            e.set_residual(y)
        for it in range(1, 13):
            so = orc.sweep(iteration=it, seed=7, **kw)
            sh = hip.sweep(iteration=it, seed=7, **kw)
            assert so["sum_delta"][0] == sh["sum_delta"][0] and so["n_events"] == sh["n_events"], f"iteration {it}"
            np.testing.assert_allclose(sh["resid_ss"], so["resid_ss"], rtol=1e-10)
            np.testing.assert_allclose(sh["alpha_ss"], so["alpha_ss"], rtol=1e-9)
        _compare(orc, hip, 1)
        # running means and X * alpha
        for e in (orc, hip):
            e.accumulate(1); e.sweep(iteration=13, seed=7, **kw); e.accumulate(2)
        for a, b in zip(orc.posterior(0), hip.posterior(0)):
            np.testing.assert_allclose(b, a, rtol=1e-9, atol=1e-12)
        np.testing.assert_allclose(hip.mul_alpha(0), orc.mul_alpha(0), rtol=1e-10, atol=1e-10)
    finally:
        hip.close()


def test_f64_bayesr_parity():
    d, X, orc, hip = _pair(600, 128 * 2 + 90, 128, "BayesR")
    try:
        y = d["y"] - d["y"].mean()
        for e in (orc, hip):
            e.set_residual(y)
            e.set_state(0, delta=np.ones(e.p, dtype=np.int32))
        kw = dict(vare=0.5, var_effect=0.05, pi_classes=np.array([0.9, 0.05, 0.03, 0.02]))
        for it in range(1, 13):
            so = orc.sweep(iteration=it, seed=9, **kw)
            sh = hip.sweep(iteration=it, seed=9, **kw)
            assert np.array_equal(so["class_counts"], sh["class_counts"]), f"iteration {it}"
            assert so["bayesr_nnz"] == sh["bayesr_nnz"]
            np.testing.assert_allclose(sh["bayesr_ssq"], so["bayesr_ssq"], rtol=1e-9)
        _compare(orc, hip, 1)
        pm = np.random.default_rng(0).dirichlet([20, 2, 1, 1], size=orc.p)          # per-marker class priors
        for it in range(13, 17):
            orc.sweep(iteration=it, seed=9, vare=0.5, var_effect=0.05, pi_matrix=pm)
            hip.sweep(iteration=it, seed=9, vare=0.5, var_effect=0.05, pi_matrix=pm)
        _compare(orc, hip, 1)
    finally:
        hip.close()


@pytest.mark.parametrize("t,bs", [(2, 128), (3, 64), (4, 128)])
def test_f64_multitrait_sampler1_parity(t, bs):
    d, X, orc, hip = _pair(500, bs * 2 + 33, bs, "MTBayesC", t=t, seed=20 + t)
    try:
        rng = np.random.default_rng(t)
        y = d["y"] - d["y"].mean()
        for k in range(t):
            yk = (1 + 0.3 * k) * y + 0.2 * rng.standard_normal(len(y))
            for e in (orc, hip):
                e.set_residual(yk, k)
                e.set_state(k, delta=np.ones(e.p))
        A = rng.standard_normal((t, t)); B = rng.standard_normal((t, t))
        kw = dict(vare=(A @ A.T / t + np.eye(t)) * 0.5, var_effect=(B @ B.T / t + np.eye(t)) * 0.003,
                  log_prior_states=np.log(rng.dirichlet(np.ones(1 << t))))
        for it in range(1, 11):
            so = orc.sweep(iteration=it, seed=4, **kw)
            sh = hip.sweep(iteration=it, seed=4, **kw)
            assert np.array_equal(so["state_counts"], sh["state_counts"]), f"iteration {it}"
            np.testing.assert_allclose(sh["beta_ss"], so["beta_ss"], rtol=1e-8)
        _compare(orc, hip, t)
    finally:
        hip.close()


@pytest.mark.parametrize("bs,nreps", [(64, 5), (128, 0)])
def test_f64_block_repetitions_parity(bs, nreps):
    """fast_blocks semantics in Float64: within-block repetitions (0 = every block its own size, BayesABC.jl:153) against
    the Float64 restatement of BayesABC_block!."""
    d, X, orc, hip = _pair(400, bs * 2 + 21, bs, "BayesC")
    try:
        y = d["y"] - d["y"].mean()
        for e in (orc, hip):
            e.set_residual(y)
        for it in range(1, 5):
            so = orc.sweep(iteration=it, seed=3, vare=0.5, var_effect=0.004, pi=0.9, nreps=nreps)
            sh = hip.sweep(iteration=it, seed=3, vare=0.5, var_effect=0.004, pi=0.9, nreps=nreps)
            assert so["sum_delta"][0] == sh["sum_delta"][0], f"iteration {it}"
        _compare(orc, hip, 1)
    finally:
        hip.close()


@pytest.mark.parametrize("method,Pi", [("BayesC", 0.9), ("BayesR", 0.0)])
def test_runmcmc_double_precision_gpu_vs_oracle(tmp_path, method, Pi):
    """runMCMC(double_precision=true) end to end: the Float64 device context against the same host loop on the Float64 oracle
    engine, identical seeds: posterior means within 1e-8."""
    d = make_dataset(n=300, p=500, ncausal=6, seed=31, center=False)
    ids = [f"i{i}" for i in range(300)]
    gdf = pd.DataFrame(d["raw"], columns=[f"m{j}" for j in range(500)])
    gdf.insert(0, "ID", ids)
    ph = pd.DataFrame({"ID": ids, "y1": d["y"]})
    outs = {}
    for tag, eng in (("orc", OracleEngine64()), ("hip", None)):
        geno = api.get_genotypes(gdf, method=method, Pi=Pi, double_precision=True)
        assert geno.genotypes.dtype == np.float64
        model = api.build_model("y1 = intercept + geno")
        outs[tag] = api.runMCMC(model, ph, chain_length=120, burnin=20, seed=11, double_precision=True,
                                output_folder=str(tmp_path / tag), _engine=eng)
    eo, eh = outs["orc"]["marker effects geno"], outs["hip"]["marker effects geno"]
    np.testing.assert_allclose(eh["Estimate"], eo["Estimate"], atol=1e-8)
    np.testing.assert_allclose(eh["Model_Frequency"], eo["Model_Frequency"], atol=1e-12)
    np.testing.assert_allclose(outs["hip"]["EBV_y1"]["EBV"], outs["orc"]["EBV_y1"]["EBV"], atol=1e-7)
    assert float(outs["hip"]["residual variance"]["Estimate"][0]) == pytest.approx(float(outs["orc"]["residual variance"]["Estimate"][0]), rel=1e-9)


# ---- round 4: the Float64 context beyond uniform 64- / 128-marker blocks -------------------
@pytest.mark.parametrize("method,bs,n,p,t", [("BayesC", 223, 600, 223 * 2 + 90, 1), ("BayesR", 512, 520, 512 + 300, 1),
                                             ("BayesB", 1024, 300, 1024 + 77, 1), ("MTBayesC", 256, 400, 256 * 2 + 31, 2)])
def test_f64_any_block_size_parity(method, bs, n, p, t):
    """Uniform blocks of ANY size <= 1024 (the reference default fast_blocks = true is floor(sqrt(n)): 223 at n = 50 000): blocks
    above 128 markers read their Gram rows from L2 and walk up to 16 sub-blocks.  Single pass = the literal chain."""
    d, X, orc, hip = _pair(n, p, bs, method, t=t, seed=41)
    try:
        rng = np.random.default_rng(5)
        y = d["y"] - d["y"].mean()
        for k in range(t):
            for e in (orc, hip):
                e.set_residual((1 + 0.4 * k) * y, k)
                if method == "MTBayesC":
                    e.set_state(k, delta=np.ones(e.p))
        if method == "BayesR":
            kw = dict(vare=0.5, var_effect=0.05, pi_classes=np.array([0.9, 0.05, 0.03, 0.02]))
        elif method == "MTBayesC":
            kw = dict(vare=np.array([[0.5, 0.1], [0.1, 0.6]]), var_effect=np.array([[0.004, 0.001], [0.001, 0.003]]),
                      log_prior_states=np.log(np.array([0.7, 0.1, 0.1, 0.1])))
        elif method == "BayesB":
            kw = dict(vare=0.5, var_effect=0.004, var_effect_vec=rng.uniform(0.002, 0.006, p), pi=0.9)
        else:
            kw = dict(vare=0.5, var_effect=0.004, pi=0.9)
        for it in range(1, 7):
            orc.sweep(iteration=it, seed=6, **kw)
            hip.sweep(iteration=it, seed=6, **kw)
        _compare(orc, hip, t)
    finally:
        hip.close()


@pytest.mark.parametrize("bs,nreps,independent", [(128, 1, False), (300, 0, False), (96, 3, True), (223, 0, True)])
def test_f64_weights_and_independent_blocks_parity(bs, nreps, independent):
    """Residual weights (x'R^-1 x, X_b'R^-1 X_b, X_b'R^-1 r, r'R^-1 r: tools4genotypes.jl:59-78,237-275) and
    independent_blocks (BayesABC.jl:190-255) in the Float64 context, against the general Float64 block oracle."""
    d, X, orc, hip = _pair(450, bs * 2 + 57, bs, "BayesC", seed=43)
    try:
        w = 1.0 + np.random.default_rng(2).uniform(0, 1.5, 450)            # Float64 weights that no Float32 holds (build_MME.jl:310)
        assert (w != w.astype(np.float32)).any()
        for e in (orc, hip):
            e.set_weights(w); e.setup_blocks(bs); e.init_state("BayesC", 1)
            e.set_residual(d["y"] - d["y"].mean())
        np.testing.assert_allclose(hip.xpx(), (X * X * w[:, None]).sum(axis=0), rtol=1e-13)       # (Float32-rounded weights would be 6e-8 off)
        for it in range(1, 6):
            so = orc.sweep(iteration=it, seed=8, vare=0.5, var_effect=0.004, pi=0.9, nreps=nreps, independent_blocks=independent)
            sh = hip.sweep(iteration=it, seed=8, vare=0.5, var_effect=0.004, pi=0.9, nreps=nreps, independent_blocks=independent)
            assert so["sum_delta"][0] == sh["sum_delta"][0], f"iteration {it}"
            np.testing.assert_allclose(sh["resid_ss"], so["resid_ss"], rtol=1e-9)
            np.testing.assert_allclose(sh["resid_sum"], so["resid_sum"], rtol=0, atol=1e-8)
        _compare(orc, hip, 1)
    finally:
        hip.close()


@pytest.mark.parametrize("t,bs,tail", [(2, 512, 126), (3, 512, 120), (2, 1024, 128)])
def test_f64_small_tail_block_beside_a_big_one(t, bs, tail):
    """A 102..128-marker tail block in a partition whose largest block is 512 / 1024 markers (p % block size; also what cutting
    an oversized block produces): the tail's Gram in LDS PLUS the per-marker arrays at the big block's stride would exceed the
    160 KB dynamic-LDS cap (b = 126, stride 512, two traits: 168 KB) -- the launch must take the Gram rows from L2 instead
    (round-4 advisor finding: the launch failed)."""
    d, X, orc, hip = _pair(260, bs + tail, bs, "MTBayesC", t=t, seed=53)
    try:
        y = d["y"] - d["y"].mean()
        for k in range(t):
            for e in (orc, hip):
                e.set_residual((1 + 0.3 * k) * y, k)
                e.set_state(k, delta=np.ones(e.p))
        R = 0.5 * np.eye(t) + 0.1
        G = 0.003 * np.eye(t) + 0.001
        lp = np.log(np.full(1 << t, 0.3 / ((1 << t) - 1))); lp[-1] = np.log(0.7)
        for it in range(1, 4):
            orc.sweep(iteration=it, seed=12, vare=R, var_effect=G, log_prior_states=lp)
            hip.sweep(iteration=it, seed=12, vare=R, var_effect=G, log_prior_states=lp)
        _compare(orc, hip, t)
    finally:
        hip.close()


def test_f64_explicit_partition_parity():
    """fast_blocks = a vector of block starts (JWAS.jl:298-304) in the Float64 context: ragged blocks, each repeated its own size."""
    d, X, orc, hip = _pair(380, 700, 128, "BayesC", seed=47)
    try:
        starts = np.array([0, 5, 140, 141, 400, 655])
        for e in (orc, hip):
            e.setup_blocks_explicit(starts); e.init_state("BayesC", 1)
            e.set_residual(d["y"] - d["y"].mean())
        for it in range(1, 4):
            so = orc.sweep(iteration=it, seed=9, vare=0.5, var_effect=0.004, pi=0.9, nreps=0)
            sh = hip.sweep(iteration=it, seed=9, vare=0.5, var_effect=0.004, pi=0.9, nreps=0)
            assert so["sum_delta"][0] == sh["sum_delta"][0], f"iteration {it}"
        _compare(orc, hip, 1)
    finally:
        hip.close()


def test_runmcmc_double_precision_default_fast_blocks_at_50000_records(tmp_path):
    """runMCMC(double_precision=true, fast_blocks=true) at n = 50 000: the reference's default block size floor(sqrt(n)) = 223
    (JWAS.jl:308-312) runs on the device (it was an error through round 3)."""
    n, p = 50_000, 1_200
    rng = np.random.default_rng(3)
    raw = rng.integers(0, 3, size=(n, p)).astype(np.float64)
    beta = np.zeros(p); beta[rng.choice(p, 10, replace=False)] = rng.standard_normal(10)
    g = (raw - raw.mean(axis=0)) @ beta
    y = 1.0 + g + rng.standard_normal(n) * g.std()
    ids = [str(i + 1) for i in range(n)]                    # (a matrix gets the individual IDs "1" .. "n", readgenotypes.jl:339-345)
    geno = api.get_genotypes(raw, method="BayesC", Pi=0.95, double_precision=True, quality_control=False)
    model = api.build_model("y1 = intercept + geno")
    ph = pd.DataFrame({"ID": ids, "y1": y})
    out = api.runMCMC(model, ph, chain_length=223 * 3, burnin=1, seed=1, double_precision=True, fast_blocks=True, outputEBV=False,
                      output_folder=str(tmp_path / "fb"))
    assert out["_timing"]["iterations"] == 3
    assert out["_timing"]["block_starts"][:3] == [1, 224, 447]
    est = np.asarray(out["marker effects geno"]["Estimate"])
    assert est.dtype == np.float64 and np.corrcoef(est, beta)[0, 1] > 0.9
