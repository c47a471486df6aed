"""End-to-end on the GPU: runMCMC through the shipped HIP engine vs the same host loop on the oracle
engine (identical seeds): posterior means within 1e-4 (the reference's own same-draws tolerance,
test/unit/test_streaming_codec.jl:100,104); plus the shard wrapper at world size 1 and full-size
properties that do not need the oracle."""
import os
import time
import numpy as np
import pandas as pd
import pytest

from conftest import make_dataset
from oracle_engine import OracleEngine
from jwas_jl_amd import api

pytestmark = pytest.mark.gpu


def _setup(method, Pi, n=400, p=1500, seed=31):
    d = make_dataset(n=n, p=p, ncausal=8, seed=seed, center=False)
    ids = [f"i{i}" for i in range(n)]
    gdf = pd.DataFrame(d["raw"], columns=[f"m{j}" for j in range(p)])
    gdf.insert(0, "ID", ids)
    ph = pd.DataFrame({"ID": ids, "y1": d["y"]})
    return gdf, ph, d


@pytest.mark.parametrize("groups", [2, 4])
def test_runmcmc_with_grouped_launches_matches_oracle_chain(tmp_path, groups):
    """runMCMC(..., blocks_per_launch=...) on a problem large enough for the adaptive block policy (512-marker blocks while many
    markers change, 1024 with grouped launches afterwards): the same host loop on the oracle engine takes the same decisions
    (they depend on the sweeps' change counts only) and the posterior means agree."""
    gdf, ph, d = _setup("BayesC", 0.99, n=300, p=4500, seed=77)
    outs, used = {}, {}
    for tag, eng in (("orc", OracleEngine("lookahead")), ("hip", None)):
        geno = api.get_genotypes(gdf, method="BayesC", Pi=0.99)
        model = api.build_model("y1 = intercept + geno")
        outs[tag] = api.runMCMC(model, ph, chain_length=120, burnin=20, seed=11, output_folder=str(tmp_path / tag),
                                _engine=eng, gram_mode="f64", blocks_per_launch=groups, outputEBV=False)
    eo = outs["orc"]["marker effects geno"]
    eh = outs["hip"]["marker effects geno"]
    np.testing.assert_allclose(eh["Estimate"], eo["Estimate"], atol=1e-4)
    np.testing.assert_allclose(eh["Model_Frequency"], eo["Model_Frequency"], atol=1e-4)
    assert float(outs["hip"]["residual variance"]["Estimate"][0]) == pytest.approx(
        float(outs["orc"]["residual variance"]["Estimate"][0]), rel=1e-4)


@pytest.mark.parametrize("method,est", [("BayesR", True), ("BayesC", False)])
def test_runmcmc_pingpong_pair_policy_matches_oracle_chain(tmp_path, method, est):
    """The host's own choice (mcmc.pingpong_pairs_for_chain: BayesR and fixed-pi chains of >= 300 iterations run their 512-marker
    sweeps as pairs with one sampler workgroup per block) -- nothing passed by the caller: the same host loop on the oracle engine
    sets the same groups up (its grouped restatement) and takes the same decisions: the same number of effect changes in every one of
    the first 100 sweeps (a chain of 450 changes per sweep at n = 300 is chaotic: the two hosts' float32-level differences in r'r flip
    a decision somewhere beyond sweep 100 of the fixed-pi chain, after which only the BayesR pair is compared on its posterior means)."""
    from jwas_jl_amd import engine as E
    gdf, ph, d = _setup(method, 0.95, n=300, p=4700, seed=78)
    outs, events, flags = {}, {}, {}
    for tag, eng in (("orc", OracleEngine("lookahead")), ("hip", None)):
        rec, grp = [], []
        events[tag], flags[tag] = rec, grp
        if eng is not None:
            orig = eng.sweep
            eng.sweep = lambda _o=orig, **kw: (lambda st: (rec.append(float(st["n_events"])), grp.append(bool(kw.get("group_launch"))), st)[2])(_o(**kw))
        else:
            orig_cls = E.HipEngine.sweep
            def sw(self, _o=orig_cls, **kw):
                st = _o(self, **kw); rec.append(float(st["n_events"])); grp.append(bool(kw.get("group_launch"))); return st
            E.HipEngine.sweep = sw
        try:
            geno = api.get_genotypes(gdf, method=method, Pi=(0.0 if method == "BayesR" else 0.95), estimatePi=est)
            model = api.build_model("y1 = intercept + geno")
            outs[tag] = api.runMCMC(model, ph, chain_length=300, burnin=50, seed=13, output_folder=str(tmp_path / tag),
                                    _engine=eng, gram_mode="f64", outputEBV=False)
        finally:
            if eng is None:
                E.HipEngine.sweep = orig_cls
        if tag == "orc":
            assert eng.blocks_per_launch(512) == 2                      # (the policy did ask for the pairs)
    assert all(flags["hip"][:100]) and all(flags["orc"][:100])          # ... and the sweeps ran them
    assert events["hip"][:100] == events["orc"][:100]
    assert min(events["hip"][:100]) > 0.0125 * 4700                     # (the high-turnover regime throughout)
    if method == "BayesR":
        eo = outs["orc"]["marker effects geno"]
        eh = outs["hip"]["marker effects geno"]
        np.testing.assert_allclose(eh["Estimate"], eo["Estimate"], atol=1e-4)
        np.testing.assert_allclose(eh["Model_Frequency"], eo["Model_Frequency"], atol=1e-4)


@pytest.mark.parametrize("method,Pi", [("BayesC", 0.95), ("BayesR", 0.0), ("BayesB", 0.9)])
def test_runmcmc_gpu_matches_oracle_chain(tmp_path, method, Pi):
    gdf, ph, d = _setup(method, Pi)
    outs = {}
    for tag, eng in (("orc", OracleEngine("lookahead")), ("hip", None)):
        geno = api.get_genotypes(gdf, method=method, Pi=Pi)
        model = api.build_model("y1 = intercept + geno")
        outs[tag] = api.runMCMC(model, ph, chain_length=200, burnin=40, seed=2026, output_folder=str(tmp_path / tag),
                                _engine=eng, block_size=256, gram_mode="f64")
    eo = outs["orc"]["marker effects geno"]
    eh = outs["hip"]["marker effects geno"]
    np.testing.assert_allclose(eh["Estimate"], eo["Estimate"], atol=1e-4)
    np.testing.assert_allclose(eh["Model_Frequency"], eo["Model_Frequency"], atol=1e-4)
    np.testing.assert_allclose(eh["SD"], eo["SD"], atol=1e-4)
    assert float(outs["hip"]["residual variance"]["Estimate"][0]) == pytest.approx(
        float(outs["orc"]["residual variance"]["Estimate"][0]), rel=1e-4)
    np.testing.assert_allclose(outs["hip"]["EBV_y1"]["EBV"], outs["orc"]["EBV_y1"]["EBV"], atol=1e-3)


def test_config1_full_chain_gpu_vs_oracle(tmp_path, config1_data):
    """BASELINE.json configs[0] / SURVEY 8d config 1: single-trait BayesC, pi0 = 0.95 estimated, 500 x 2000, 1000
    iterations, burn-in 100 -- the whole chain on the device against the oracle-driven chain, same seed.
    Tier-1 tolerance: |d posterior mean alpha| <= 1e-4, |d sigma2_e| / sigma2_e <= 1e-4."""
    d = config1_data
    n, p = d["X"].shape
    ids = [f"i{i}" for i in range(n)]
    gdf = pd.DataFrame(d["raw"], columns=[f"m{j}" for j in range(p)])
    gdf.insert(0, "ID", ids)
    ph = pd.DataFrame({"ID": ids, "y1": d["y"]})
    outs = {}
    for tag, eng in (("orc", OracleEngine("lookahead")), ("hip", None)):
        geno = api.get_genotypes(gdf, method="BayesC", Pi=0.95, estimatePi=True)
        model = api.build_model("y1 = intercept + geno")
        outs[tag] = api.runMCMC(model, ph, chain_length=1000, burnin=100, seed=2026, outputEBV=False,
                                output_folder=str(tmp_path / tag), _engine=eng, block_size=256, gram_mode="f64")
    eo, eh = outs["orc"]["marker effects geno"], outs["hip"]["marker effects geno"]
    assert np.abs(eh["Estimate"] - eo["Estimate"]).max() <= 1e-4
    assert np.abs(eh["Model_Frequency"] - eo["Model_Frequency"]).max() <= 1e-4
    ro, rh = (float(outs[k]["residual variance"]["Estimate"][0]) for k in ("orc", "hip"))
    assert abs(rh - ro) / ro <= 1e-4
    po, phh = (float(outs[k]["pi_geno"]["Estimate"][0]) for k in ("orc", "hip"))
    assert abs(phh - po) <= 1e-4
    causal = {f"m{j}" for j in d["causal"]}
    top = set(eh.reindex(eh["Model_Frequency"].sort_values(ascending=False).index)["Marker_ID"].head(20))
    assert len(top & causal) >= 8


def test_three_trait_chain_gpu_vs_oracle(tmp_path):
    """Config 4 shape (3-trait BayesC, sampler I, residual covariance drawn on the host), scaled down."""
    d = make_dataset(n=300, p=450, ncausal=6, seed=77, center=False)
    rng = np.random.default_rng(4)
    ids = [f"i{i}" for i in range(300)]
    gdf = pd.DataFrame(d["raw"], columns=[f"m{j}" for j in range(450)])
    gdf.insert(0, "ID", ids)
    ph = pd.DataFrame({"ID": ids, "y1": d["y"], "y2": (0.7 * d["y"] + 0.5 * rng.standard_normal(300)).astype(np.float32),
                       "y3": (-0.4 * d["y"] + 0.8 * rng.standard_normal(300)).astype(np.float32)})
    Pi = {tuple(float(b) for b in f"{s:03b}"): pr for s, pr in enumerate([0.6, 0.05, 0.05, 0.05, 0.05, 0.05, 0.05, 0.1])}
    outs = {}
    for tag, eng in (("orc", OracleEngine("lookahead")), ("hip", None)):
        geno = api.get_genotypes(gdf, method="BayesC", Pi=Pi, estimatePi=True)
        model = api.build_model("y1 = intercept + geno\ny2 = intercept + geno\ny3 = intercept + geno")
        outs[tag] = api.runMCMC(model, ph, chain_length=80, burnin=10, seed=11, outputEBV=False,
                                output_folder=str(tmp_path / tag), _engine=eng, block_size=128, gram_mode="f64")
    eo, eh = outs["orc"]["marker effects geno"], outs["hip"]["marker effects geno"]
    assert len(eh) == 3 * 450
    np.testing.assert_allclose(eh["Estimate"], eo["Estimate"], atol=1e-4)
    np.testing.assert_allclose(eh["Model_Frequency"], eo["Model_Frequency"], atol=1e-4)
    np.testing.assert_allclose(outs["hip"]["residual variance"]["Estimate"], outs["orc"]["residual variance"]["Estimate"], rtol=1e-3)


@pytest.mark.parametrize("method", ["BayesB", "BayesA"])
def test_three_trait_bayesb_chain_gpu_vs_oracle(tmp_path, method):
    """Multi-trait BayesA/B through runMCMC: every marker's 3 x 3 effect covariance is redrawn on the host each iteration
    (variance_components.jl:181-186) and inverted on the device; the whole chain against the CPU oracle engine."""
    d = make_dataset(n=280, p=400, ncausal=6, seed=78, center=False)
    rng = np.random.default_rng(5)
    ids = [f"i{i}" for i in range(280)]
    gdf = pd.DataFrame(d["raw"], columns=[f"m{j}" for j in range(400)])
    gdf.insert(0, "ID", ids)
    ph = pd.DataFrame({"ID": ids, "y1": d["y"], "y2": (0.7 * d["y"] + 0.5 * rng.standard_normal(280)).astype(np.float32),
                       "y3": (-0.4 * d["y"] + 0.8 * rng.standard_normal(280)).astype(np.float32)})
    outs = {}
    for tag, eng in (("orc", OracleEngine("lookahead")), ("hip", None)):
        geno = api.get_genotypes(gdf, method=method)
        model = api.build_model("y1 = intercept + geno\ny2 = intercept + geno\ny3 = intercept + geno")
        outs[tag] = api.runMCMC(model, ph, chain_length=60, burnin=10, seed=12, outputEBV=False,
                                output_folder=str(tmp_path / tag), _engine=eng, block_size=128, gram_mode="f64")
    eo, eh = outs["orc"]["marker effects geno"], outs["hip"]["marker effects geno"]
    assert len(eh) == 3 * 400
    np.testing.assert_allclose(eh["Estimate"], eo["Estimate"], atol=1e-4)
    np.testing.assert_allclose(eh["Model_Frequency"], eo["Model_Frequency"], atol=1e-4)
    np.testing.assert_allclose(outs["hip"]["residual variance"]["Estimate"], outs["orc"]["residual variance"]["Estimate"], rtol=1e-3)


@pytest.mark.parametrize("method,Pi", [("BayesC", 0.9), ("BayesR", 0.0)])
def test_runmcmc_non_uniform_fast_blocks_gpu_vs_oracle(tmp_path, method, Pi):
    """fast_blocks = a vector of block starts with blocks of different sizes (JWAS.jl:298-304): the device runs exactly that
    partition (jwas_hip_setup_blocks_explicit), every block its own size as repetition count; chain_length is not rescaled."""
    d = make_dataset(n=300, p=520, ncausal=8, seed=31, center=False)
    ids = [f"i{i}" for i in range(300)]
    gdf = pd.DataFrame(d["raw"], columns=[f"m{j}" for j in range(520)])
    gdf.insert(0, "ID", ids)
    ph = pd.DataFrame({"ID": ids, "y": d["y"]})
    # 1-based, as in the reference; blocks of 1..40 markers (a block runs its own size as repetition count, so the chain
    # makes up to 40 x chain_length passes over a marker: kept short, the comparison is float-for-float)
    starts = [1, 31, 32, 70, 110, 150, 185, 225, 260, 300, 340, 380, 420, 460, 500]
    outs = {}
    for tag, eng in (("orc", OracleEngine("lookahead")), ("hip", None)):
        geno = api.get_genotypes(gdf, method=method, Pi=Pi, estimatePi=True)
        model = api.build_model("y = intercept + geno")
        outs[tag] = api.runMCMC(model, ph, chain_length=10, burnin=2, seed=5, outputEBV=False, fast_blocks=starts,
                                output_folder=str(tmp_path / tag), _engine=eng, gram_mode="f64")
        assert outs[tag]["_timing"]["iterations"] == 10
    eo, eh = outs["orc"]["marker effects geno"], outs["hip"]["marker effects geno"]
    np.testing.assert_allclose(eh["Estimate"], eo["Estimate"], atol=1e-4)
    np.testing.assert_allclose(eh["Model_Frequency"], eo["Model_Frequency"], atol=1e-4)
    np.testing.assert_allclose(outs["hip"]["residual variance"]["Estimate"], outs["orc"]["residual variance"]["Estimate"], rtol=1e-3)


def test_runmcmc_gpu_mfma_gram_statistically_equivalent(tmp_path):
    """With the production (fp32 MFMA) Gram the chain may round differently from the oracle; the
    posterior summaries still agree within Monte-Carlo noise of a short chain."""
    gdf, ph, d = _setup("BayesC", 0.95)
    geno = api.get_genotypes(gdf, method="BayesC", Pi=0.95)
    model = api.build_model("y1 = intercept + geno")
    out = api.runMCMC(model, ph, chain_length=300, burnin=50, seed=2026, output_folder=str(tmp_path / "mfma"), block_size=256)
    me = out["marker effects geno"]
    causal = {f"m{j}" for j in d["causal"]}
    top = set(me.reindex(me["Model_Frequency"].sort_values(ascending=False).index)["Marker_ID"].head(10))
    assert len(top & causal) >= 4
    assert np.corrcoef(out["EBV_y1"]["EBV"], ph["y1"])[0, 1] > 0.5


def test_marker_shard_world1_is_the_plain_sweep():
    import jwas_jl_amd as J
    from jwas_jl_amd.dist import MarkerShard
    d = make_dataset(n=500, p=700, ncausal=5, seed=8)
    r0 = (d["y"] - d["y"].mean()).astype(np.float32)
    outs = []
    for use_shard in (False, True):
        e = J.HipEngine(0)
        e.load_dense(d["X"]); e.setup_blocks(256, "f64"); e.init_state("BayesC")
        r = r0[None, :].copy()
        sh = MarkerShard(e, 0, 700)
        for it in range(1, 4):
            if use_shard:
                r, st = sh.sweep(r, iteration=it, seed=3, vare=np.float32(0.5), var_effect=np.float32(0.004), pi=0.9)
            else:
                e.set_residual(r[0]); st = e.sweep(iteration=it, seed=3, vare=np.float32(0.5), var_effect=np.float32(0.004), pi=0.9)
                r = e.get_residual()[None, :]
        outs.append((e.get_state()[0], r.copy(), st["resid_ss"]))
        e.close()
    assert np.array_equal(outs[0][0], outs[1][0]) and np.array_equal(outs[0][1], outs[1][1])


def test_larger_problem_invariants_without_oracle():
    """Size-independent properties at a size the oracle would take minutes for: the residual identity
    r = y - X alpha holds after many sweeps, statistics equal direct reductions of the state, the
    chain is reproducible, and block size does not change the draws' outcome beyond fp32 noise."""
    import jwas_jl_amd as J
    n, p = 3000, 20000
    res = {}
    for bs in (256, 1024):
        e = J.HipEngine(0)
        e.alloc_dense(n, p); e.synth(11, 0, True); e.setup_blocks(bs, "mfma"); e.init_state("BayesC")
        rng = np.random.default_rng(0)
        a_true = np.zeros(p, dtype=np.float32); idx = rng.choice(p, 20, replace=False); a_true[idx] = rng.standard_normal(20)
        e.set_state(alpha=a_true)
        g = e.mul_alpha()
        y = (g / g.std() + rng.standard_normal(n)).astype(np.float32)
        e.set_state(alpha=np.zeros(p), beta=np.zeros(p), delta=np.zeros(p))
        e.set_residual(y - y.mean())
        for it in range(1, 11):
            st = e.sweep(iteration=it, seed=99, vare=np.float32(1.0), var_effect=np.float32(0.01), pi=0.99)
        a, b, dlt = e.get_state()
        r = e.get_residual()
        np.testing.assert_allclose(r, (y - y.mean()) - e.mul_alpha(), atol=2e-3)          # residual identity
        assert st["sum_delta"][0] == float(dlt.sum()) == float((a != 0).sum())
        assert st["alpha_ss"][0, 0] == pytest.approx(float(a.astype(np.float64) @ a.astype(np.float64)), rel=1e-9)
        assert st["resid_ss"][0, 0] == pytest.approx(float(r.astype(np.float64) @ r.astype(np.float64)), rel=1e-9)
        assert set(np.flatnonzero(np.abs(a) > 0.2)) <= set(idx) | set(np.flatnonzero(dlt))   # strong effects are real or flagged
        res[bs] = (a, dlt)
        e.close()
    same = (res[256][1] == res[1024][1]).mean()
    assert same > 0.999                      # identical draws; only fp32 rounding of the two Gram layouts differs


def test_full_size_config2_invariants():
    """BASELINE.json config 2 at its full size (50 000 x 600 000 fp32, 120 GB in HBM, genotypes generated on the
    device): the oracle cannot run here, so parity is checked through size-independent properties -- the residual
    identity r = y - X alpha after several sweeps, statistics = direct reductions of the state, bit-reproducibility,
    and invariance of the chain to the block size (512 vs 1024: same draws; only the fp32 rounding of the two Gram
    layouts differs)."""
    import jwas_jl_amd as J
    n, p = 50_000, 600_000
    e = J.HipEngine(0)
    if e.device_info()["hbm_free"] < 150e9:
        e.close()
        pytest.skip("needs 150 GB of free HBM")
    e.alloc_dense(n, p); e.synth(2026, 0, True)
    e.setup_blocks(512, "mfma"); e.add_block_size(1024, "mfma")
    e.init_state("BayesC")
    rng = np.random.default_rng(0)
    a_true = np.zeros(p, dtype=np.float32); idx = rng.choice(p, 600, replace=False); a_true[idx] = rng.standard_normal(600)
    e.set_state(alpha=a_true)
    g = e.mul_alpha()
    y = (g / g.std() + rng.standard_normal(n)).astype(np.float32)
    y -= y.mean()
    xpx = e.xpx()
    assert xpx.min() > 0 and np.isfinite(xpx).all()
    varg = np.float32(1.0 / (0.05 * float(xpx.astype(np.float64).sum()) / n))
    res = {}
    # ("d" .. "f": GROUPED LAUNCHES on the 1024-marker blocks -- 4 blocks per launch, twice, then 2: jwas_hip_setup_groups)
    for tag, bs, m in (("a", 512, 0), ("b", 1024, 0), ("c", 512, 0), ("d", 1024, 4), ("e", 1024, 4), ("f", 1024, 2)):
        e.select_block_size(bs)
        if m and e.blocks_per_launch() != m:
            e.setup_groups(m, "mfma")
        e.set_state(alpha=np.zeros(p), beta=np.zeros(p), delta=np.ones(p))
        e.set_residual(y)
        pi = 0.95
        for it in range(1, 7):
            st = e.sweep(iteration=it, seed=2026, vare=np.float32(1.0), var_effect=varg, pi=pi, group_launch=bool(m))
            pi = float(1 - (st["sum_delta"][0] + 1) / (p + 2))
        a, b, dlt = e.get_state()
        r = e.get_residual()
        np.testing.assert_allclose(r, y - e.mul_alpha(), atol=5e-3)                              # residual identity
        assert st["sum_delta"][0] == float(dlt.sum()) == float((a != 0).sum())
        assert st["alpha_ss"][0, 0] == pytest.approx(float(a.astype(np.float64) @ a.astype(np.float64)), rel=1e-9)
        assert st["resid_ss"][0, 0] == pytest.approx(float(r.astype(np.float64) @ r.astype(np.float64)), rel=1e-9)
        res[tag] = (a, dlt, r)
    e.close()
    # the same genotypes kept 2-bit packed in HBM (7.5 GB instead of 120 GB): the same chain up to the rounding of the packed
    # update role's right-hand sides (its own summation order: the centring is factored out of the sum, update_role.hpp)
    e = J.HipEngine(0)
    e.alloc_packed(n, p); e.synth(2026, 0, True)
    e.setup_blocks(512, "mfma"); e.init_state("BayesC")
    assert np.array_equal(e.xpx(), xpx)
    e.set_state(alpha=np.zeros(p), beta=np.zeros(p), delta=np.ones(p))
    e.set_residual(y)
    pi = 0.95
    for it in range(1, 7):
        st = e.sweep(iteration=it, seed=2026, vare=np.float32(1.0), var_effect=varg, pi=pi)
        pi = float(1 - (st["sum_delta"][0] + 1) / (p + 2))
    res["packed"] = (e.get_state()[0], e.get_state()[2], e.get_residual())
    e.close()
    assert (res["packed"][1] == res["a"][1]).mean() > 0.998            # the same draws: indicators agree but for roundings amplified over 6 cold-start sweeps
    bothp = (res["packed"][1] != 0) & (res["a"][1] != 0)
    assert np.abs(res["packed"][0][bothp] - res["a"][0][bothp]).max() < 5e-3
    assert np.array_equal(res["a"][0], res["c"][0]) and np.array_equal(res["a"][2], res["c"][2])    # reproducible, bit for bit
    assert (res["a"][1] == res["b"][1]).mean() > 0.998                                          # block-size invariant draws (the chains differ only by the fp32 rounding of the two Gram layouts, amplified over 6 cold-start sweeps)
    both = (res["a"][1] != 0) & (res["b"][1] != 0)
    assert np.abs(res["a"][0][both] - res["b"][0][both]).max() < 5e-3
    # grouped launches: bit-reproducible, and the plain 1024-marker chain up to the rounding of the regrouped right-hand sides
    assert np.array_equal(res["d"][0], res["e"][0]) and np.array_equal(res["d"][2], res["e"][2])
    for tag in ("d", "f"):
        assert (res[tag][1] == res["b"][1]).mean() > 0.998
        bothg = (res[tag][1] != 0) & (res["b"][1] != 0)
        assert np.abs(res[tag][0][bothg] - res["b"][0][bothg]).max() < 5e-3
    # the simulated QTL with large effects are found
    big = idx[np.abs(a_true[idx]) > 1.5]
    assert (res["a"][1][big] != 0).mean() > 0.5


def test_marker_shard_with_nccl_backend_single_rank(tmp_path):
    """The collective path (RCCL all-reduce of the residual delta and of the packed statistics) with the real
    nccl backend: one rank, collectives forced, must reproduce the plain sweep up to the fp32 rounding of
    snapshot + (local - snapshot)."""
    import subprocess
    import sys
    import os
    script = tmp_path / "nccl_one_rank.py"
    script.write_text("""
import os, sys
import numpy as np
sys.path[:0] = [os.environ["REPO"], os.path.join(os.environ["REPO"], "tests")]
import torch, torch.distributed as dist
from conftest import make_dataset
import jwas_jl_amd as J
from jwas_jl_amd.dist import MarkerShard
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1)
d = make_dataset(n=500, p=700, ncausal=5, seed=8)
r0 = (d["y"] - d["y"].mean()).astype(np.float32)
outs = []
for force in (False, True):
    e = J.HipEngine(0); e.load_dense(d["X"]); e.setup_blocks(256, "f64"); e.init_state("BayesC")
    sh = MarkerShard(e, 0, 700, 0, 1, force_collective=force)
    r = r0[None, :].copy()
    for it in range(1, 4):
        r, st = sh.sweep(r, iteration=it, seed=3, vare=np.float32(0.5), var_effect=np.float32(0.004), pi=0.9)
    outs.append((e.get_state()[0], r.copy(), float(st["resid_ss"][0, 0]), float(st["sum_delta"][0])))
    e.close()
# r_snapshot + (r_local - r_snapshot) differs from r_local by fp32 rounding (<= 1 ulp per sweep)
assert np.abs(outs[0][0] - outs[1][0]).max() <= 1e-5 and np.abs(outs[0][1] - outs[1][1]).max() <= 1e-5
assert abs(outs[0][2] - outs[1][2]) <= 1e-5 * abs(outs[0][2]) and outs[0][3] == outs[1][3]
dist.destroy_process_group()
print("NCCL_ONE_RANK_OK")
""")
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29533", REPO=os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
               HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run([sys.executable, str(script)], env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "NCCL_ONE_RANK_OK" in out.stdout, out.stdout[-1500:] + out.stderr[-1500:]


@pytest.mark.parametrize("groups", [0, 2])
def test_library_sharded_sweep_single_rank(groups):
    """jwas_hip_comm_init / jwas_hip_sweep_sharded through the C ABI with a one-rank RCCL communicator: the on-device
    reconcile (pack kernel, ncclAllReduce, apply kernel) must reproduce the plain sweep up to the fp32 rounding of
    snapshot + (local - snapshot), with identical statistics.  groups = 2: both sides with grouped launches (the shard's own
    sweep is the same launch sequence, jwas_sweep_params.group_launch)."""
    import jwas_jl_amd as J
    d = make_dataset(n=700, p=900, ncausal=6, seed=18)
    r0 = (d["y"] - d["y"].mean()).astype(np.float32)
    outs = []
    for sharded in (False, True):
        e = J.HipEngine(0)
        e.load_dense(d["X"]); e.setup_blocks(256 if not groups else 128, "f64"); e.init_state("BayesC")
        if groups:
            e.setup_groups(groups, "f64")
        e.set_residual(r0)
        if sharded:
            e.comm_init(J.HipEngine.comm_unique_id(), 0, 1)
        for it in range(1, 5):
            fn = e.sweep_sharded if sharded else e.sweep
            st = fn(iteration=it, seed=3, vare=np.float32(0.5), var_effect=np.float32(0.004), pi=0.9, group_launch=bool(groups))
        outs.append((e.get_state()[0], e.get_residual(), st))
        e.close()
    (a0, r0_, s0), (a1, r1_, s1) = outs
    assert np.abs(a0 - a1).max() <= 1e-5 and np.abs(r0_ - r1_).max() <= 1e-5
    assert s0["sum_delta"][0] == s1["sum_delta"][0] and s0["n_events"] == s1["n_events"]
    assert s1["alpha_ss"][0, 0] == pytest.approx(s0["alpha_ss"][0, 0], rel=1e-4)
    assert s1["resid_ss"][0, 0] == pytest.approx(float(r1_.astype(np.float64) @ r1_.astype(np.float64)), rel=1e-9)


def test_two_rank_sharded_sweep_over_rccl(tmp_path):
    """Two processes, two GPUs, the library's sharded sweep over RCCL (skipped on a one-GPU box): must equal the
    single-process emulation -- two contexts on one GPU, delta r summed in fp64 on the host (a two-term sum does not
    depend on the order) -- bit for bit."""
    import subprocess
    import sys
    import os
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs")
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", REPO=repo)
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                          "--master-port", "29541", os.path.join(repo, "tests", "_dist_gpu_worker.py")], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "TWO_RANK_RCCL_OK" in out.stdout, out.stdout[-2000:] + out.stderr[-2000:]


def test_two_rank_row_shards_over_rccl(tmp_path):
    """Two processes, two GPUs, exact row shards over RCCL (skipped on a one-GPU box): the same chain, bit for bit, as the
    same two shards over the loopback transport on one GPU."""
    import subprocess
    import sys
    import os
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs")
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", REPO=repo)
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                          "--master-port", "29543", os.path.join(repo, "tests", "_dist_gpu_rows_worker.py")], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "TWO_RANK_ROWS_RCCL_OK" in out.stdout, out.stdout[-2000:] + out.stderr[-2000:]


def test_heldout_ebv_gpu_matches_oracle(tmp_path):
    """Individuals with genotypes but no record: EBV = output_genotypes * alpha on the device
    (jwas_hip_load_output_dense_f32 / jwas_hip_mul_alpha_output; output.jl:281-306, tools4genotypes.jl:290-296)."""
    gdf, ph, d = _setup("BayesC", 0.95, n=330, p=700, seed=5)
    ph = ph.copy()
    held = np.arange(3, 330, 4)
    ph.loc[held, "y1"] = np.nan
    outs = {}
    for tag, eng in (("orc", OracleEngine("lookahead")), ("hip", None)):
        geno = api.get_genotypes(gdf, method="BayesC", Pi=0.95)
        model = api.build_model("y1 = intercept + geno")
        outs[tag] = api.runMCMC(model, ph, chain_length=120, burnin=20, seed=9, output_folder=str(tmp_path / tag),
                                _engine=eng, block_size=128, gram_mode="f64")
    eo, eh = outs["orc"]["EBV_y1"], outs["hip"]["EBV_y1"]
    assert list(eh["ID"]) == list(eo["ID"]) == list(gdf["ID"])
    np.testing.assert_allclose(eh["EBV"], eo["EBV"], atol=1e-3)
    np.testing.assert_allclose(eh["PEV"], eo["PEV"], atol=1e-3)
    assert np.corrcoef(eh["EBV"].to_numpy()[held], d["y"][held])[0, 1] > 0.3


def test_reference_cv_benchmark_heldout_accuracy(tmp_path):
    """Statistical pin against the reference's OWN published output: its 5-fold cross-validation benchmark on its
    packaged simulated_annotations data (benchmarks/simulated_annotations_multitrait_comparison.jl, cv mode; results in
    benchmarks/reports/2026-04-11-simulated-annotations-cv-report.md): held-out cor(y, EBV), trait mean, BayesC_single
    0.6424 and MT_BayesC_I 0.6397 (2 seeds x 5 folds).  Same protocol through this package, one seed here (the full
    two-seed run: scripts/cv_simulated_annotations.py -> profiles/r01_cv_simulated_annotations.json: 0.6434 / 0.6394;
    standard error of a fold mean ~0.015, fold partitions differ from the reference's)."""
    import importlib.util
    import os
    spec = importlib.util.spec_from_file_location(
        "cv_sa", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "scripts", "cv_simulated_annotations.py"))
    cv = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(cv)
    pheno = pd.read_csv(os.path.join(cv.DATA, "phenotypes_mt.csv"), dtype={"ID": str})
    fold_of = cv.folds_for(list(pheno["ID"]), 5, 101)
    rows = []
    for case in ("BayesC_y1", "BayesC_y2", "MT_I"):
        for fold in range(1, 6):
            rows += cv.run_variant(case, pheno, 101, fold_of, fold, 1500, 500, 50, str(tmp_path))
    df = pd.DataFrame(rows)
    df["family"] = df["variant"].str.replace("_y1", "").str.replace("_y2", "")
    got = {fam: float(g.groupby("trait")["cor"].mean().mean()) for fam, g in df.groupby("family")}
    assert abs(got["BayesC"] - cv.REFERENCE["BayesC"]) < 0.04, got
    assert abs(got["MT_I"] - cv.REFERENCE["MT_I"]) < 0.04, got


def test_impute_genotypes_on_device_and_sweep_parity():
    """Single-step input assembled chunk by chunk straight into HBM (impute_genotypes, SSBR.jl:83-142: real-valued rows
    for the non-genotyped individuals): the device matrix equals the host assembly, and a BayesC chain on it equals the
    oracle's chain on the host matrix (indicator trajectories bit for bit)."""
    import pandas as pd
    import jwas_jl_amd as J
    from jwas_jl_amd import api, single_step as SS
    from test_single_step import _random_pedigree
    ped = SS.get_pedigree(_random_pedigree(20, 500, seed=9))
    rng = np.random.default_rng(4)
    genotyped = sorted(rng.choice(ped.ids, 160, replace=False))
    d = make_dataset(n=160, p=700, ncausal=8, seed=12, center=False)
    gdf = pd.DataFrame(d["raw"], columns=[f"m{j}" for j in range(700)]); gdf.insert(0, "ID", genotyped)
    geno = api.get_genotypes(gdf, 1.0, method="BayesC", Pi=0.9)
    pheno_ids = list(rng.permutation(ped.ids)[:450])
    host = SS.impute_genotypes(geno, ped, pheno_ids, markers_per_chunk=128, return_host=True)
    e = J.HipEngine(0)
    dev = SS.impute_genotypes(geno, ped, pheno_ids, engine=e, markers_per_chunk=128)
    assert dev.storage_mode == "device" and (e.n, e.p) == host.genotypes.shape
    assert np.array_equal(e.get_columns(0, e.p), host.genotypes)
    X = np.asfortranarray(host.genotypes)
    y = (X[:, :5].astype(np.float64) @ rng.standard_normal(5) + rng.standard_normal(e.n)).astype(np.float32)
    y -= y.mean()
    orc = OracleEngine("lookahead")
    orc.load_dense(X); orc.setup_blocks(256); orc.init_state("BayesC")
    e.setup_blocks(256, "f64"); e.init_state("BayesC")
    for eng in (orc, e):
        eng.set_residual(y)
    for it in range(1, 9):
        kw = dict(iteration=it, seed=2, vare=np.float32(1.0), var_effect=np.float32(0.05), pi=0.9)
        orc.sweep(**kw); e.sweep(**kw)
    ao, _, do = orc.get_state()
    ah, _, dh = e.get_state()
    assert np.array_equal(do, dh) and do.sum() > 3
    np.testing.assert_allclose(ah, ao, rtol=0, atol=5e-6)
    e.close()


@pytest.mark.parametrize("method,t,bs,nranks", [("BayesC", 1, 128, 2), ("BayesR", 1, 256, 2), ("MTBayesC", 2, 128, 3), ("BayesC", 1, 64, 4)])
def test_exact_row_shards_over_the_loopback_transport(method, t, bs, nranks):
    """Exact ROW shards (jwas_hip_comm_row_shards): every rank holds a slice of the individuals and all markers; x'x, the Grams
    and each block's partial RHS are summed over the ranks, the sampler runs replicated.  The ranks here are engines of one
    process on different host threads (loopback transport: the all-reduces go through host memory), which exercises the same
    call sites the RCCL transport uses.  The pooled chain must equal the oracle's on the full data (the sums are formed in
    a different order, nothing else), every rank must hold the same effects bit for bit, and r'r must be the pooled one."""
    import threading
    import jwas_jl_amd as J
    rows_per = 512                                       # two 256-row slices per rank: the same row groups everywhere
    n = rows_per * nranks
    d = make_dataset(n=n, p=3 * bs + 21, ncausal=8, seed=50 + nranks)
    p = d["X"].shape[1]
    y = (d["y"] - d["y"].mean()).astype(np.float32)
    vare, varg = np.float32(0.5 * d["y"].var()), np.float32(0.004)
    if method == "BayesR":
        kw = dict(vare=vare, var_effect=np.float32(0.05), pi_classes=np.array([0.9, 0.05, 0.03, 0.02]))
    elif t == 1:
        kw = dict(vare=vare, var_effect=varg, pi=0.9)
    else:
        kw = dict(vare=(np.eye(t) * 0.5 + 0.1).astype(np.float32) * vare, var_effect=(np.eye(t) * 0.004).astype(np.float32),
                  log_prior_states=np.log(np.array([0.7, 0.1, 0.1, 0.1])))
    nsweeps = 6
    orc = OracleEngine("lookahead")
    orc.load_dense(d["X"]); orc.setup_blocks(bs); orc.init_state(method, t)
    for k in range(t):
        orc.set_residual(((1 + 0.3 * k) * y).astype(np.float32), k)
        if t > 1:
            orc.set_state(k, delta=np.ones(p, dtype=np.float32))
    so = [orc.sweep(iteration=it, seed=23, **kw) for it in range(1, nsweeps + 1)]
    out, errs = [None] * nranks, []

    def run(rank):
        try:
            e = J.HipEngine(0)
            sl = slice(rank * rows_per, (rank + 1) * rows_per)
            e.load_dense(np.asfortranarray(d["X"][sl]))
            e.comm_init_loopback(1, rank, nranks)
            e.comm_row_shards(True)
            e.setup_blocks(bs, "f64"); e.init_state(method, t)
            for k in range(t):
                e.set_residual(((1 + 0.3 * k) * y[sl]).astype(np.float32), k)
                if t > 1:
                    e.set_state(k, delta=np.ones(p, dtype=np.float32))
            st = [e.sweep(iteration=it, seed=23, **kw) for it in range(1, nsweeps + 1)]
            out[rank] = ([e.get_state(k) for k in range(t)], [e.get_residual(k) for k in range(t)], st, e.xpx())
            e.close()
        except Exception as ex:                          # noqa: BLE001
            errs.append((rank, repr(ex)))

    th = [threading.Thread(target=run, args=(r,)) for r in range(nranks)]
    for x in th:
        x.start()
    for x in th:
        x.join(timeout=300)
    assert not errs, errs
    assert all(o is not None for o in out)
    np.testing.assert_allclose(out[0][3], orc.xpx(), rtol=2e-6)                      # x'x of the pooled individuals
    for it in range(nsweeps):
        for r in range(nranks):
            assert out[r][2][it]["n_events"] == so[it]["n_events"], f"sweep {it + 1}, rank {r}"
            np.testing.assert_allclose(out[r][2][it]["resid_ss"], so[it]["resid_ss"], rtol=2e-5)      # pooled r'r on every rank
    for k in range(t):
        ao, bo, do = orc.get_state(k)
        for r in range(nranks):
            a, b_, dl = out[r][0][k]
            assert np.array_equal(a, out[0][0][k][0]) and np.array_equal(dl, out[0][0][k][2])      # replicated bit for bit
            assert np.array_equal(dl, do)
            np.testing.assert_allclose(a, ao, rtol=0, atol=2e-5)
        r_all = np.concatenate([out[r][1][k] for r in range(nranks)])
        np.testing.assert_allclose(r_all, orc.get_residual(k), rtol=0, atol=1e-4)


def test_row_shards_single_rank_is_the_plain_sweep_and_contract_errors():
    import jwas_jl_amd as J
    d = make_dataset(n=300, p=200, ncausal=5, seed=8)
    y = (d["y"] - d["y"].mean()).astype(np.float32)
    kw = dict(vare=np.float32(0.5), var_effect=np.float32(0.004), pi=0.9)
    res = []
    for row in (False, True, "rccl"):
        e = J.HipEngine(0)
        e.load_dense(d["X"])
        if row == "rccl":                                # the RCCL transport with one rank: same call sites, real ncclAllReduce
            # (ncclCommInitRank now and then fails with "unhandled cuda error" while OTHER processes are bringing contexts up on the
            # same GPU -- pytest -n 4 on a one-GPU box; nothing of ours has run on the communicator yet: try the bring-up again)
            for attempt in range(4):
                try:
                    e.comm_init(e.comm_unique_id(), 0, 1)
                    break
                except J.JwasHipError as err:
                    if "unhandled cuda error" not in str(err):
                        raise
                    if attempt == 3:
                        # (only ever seen inside pytest -n 4 runs, where it then persists for the process; a serial run -- the driver's
                        # -- must pass for real)
                        if os.environ.get("PYTEST_XDIST_WORKER"):
                            pytest.skip("RCCL bring-up keeps failing while other worker processes share the GPU")
                        raise
                    e.close()
                    time.sleep(1.0 + attempt)
                    e = J.HipEngine(0)
                    e.load_dense(d["X"])
            e.comm_row_shards(True)
        elif row:
            with pytest.raises(J.JwasHipError, match="attach a communicator first"):
                e.comm_row_shards(True)
            e.comm_init_loopback(2, 0, 1)
            e.comm_row_shards(True)
        e.setup_blocks(64, "f64"); e.init_state("BayesC"); e.set_residual(y)
        st = [e.sweep(iteration=it, seed=4, **kw) for it in range(1, 5)]
        res.append((e.get_state(), e.get_residual(), st[-1]["resid_ss"]))
        if row is True:
            with pytest.raises(J.JwasHipError, match="before jwas_hip_setup_blocks"):
                e.comm_row_shards(False)
            with pytest.raises(J.JwasHipError, match="not available on row shards"):
                e.sweep(iteration=9, seed=4, independent_blocks=True, **kw)
        e.close()
    for q in (1, 2):
        assert all(np.array_equal(x, y_) for x, y_ in zip(res[0][0], res[q][0])) and np.array_equal(res[0][1], res[q][1])
        assert np.array_equal(res[0][2], res[q][2])


def test_device_resident_engine_is_reset_between_runs(tmp_path):
    """A device-resident engine (api.device_genotypes) reused across runMCMC calls: a WEIGHTED run followed by an
    unweighted one, and a run on an explicit block partition followed by a plain one, must give exactly what a fresh
    engine gives -- the previous run's R^-1, its weighted Grams and its partition may not leak into the next run."""
    from jwas_jl_amd.engine import HipEngine
    d = make_dataset(n=300, p=900, ncausal=6, seed=77)
    ids = [str(i + 1) for i in range(300)]
    rng = np.random.default_rng(5)
    wts = rng.uniform(0.5, 2.0, 300)
    ph = pd.DataFrame({"ID": ids, "y1": d["y"], "weights": wts})

    info = {}

    def run(eng, tag, **kw):
        # (allele frequencies / sum 2pq come from the unweighted x'x of the first call; a weighted engine refuses to derive them)
        geno = api.device_genotypes(eng, method="BayesC", Pi=0.9, estimatePi=True, obsID=ids, **info)
        info.update(alleleFreq=geno.alleleFreq, sum2pq=geno.sum2pq)
        model = api.build_model("y1 = intercept + geno", genotypes={"geno": geno})
        return api.runMCMC(model, ph, chain_length=30, burnin=5, seed=4, outputEBV=False, output_folder=str(tmp_path / tag),
                           printout_model_info=False, **kw)["marker effects geno"]

    fresh = HipEngine(0)
    fresh.load_dense(d["X"])
    ref = run(fresh, "fresh")
    fresh.close()

    eng = HipEngine(0)
    eng.load_dense(d["X"])
    run(eng, "weighted", heterogeneous_residuals=True)
    assert eng._weighted
    again = run(eng, "after_weighted")
    assert not eng._weighted
    np.testing.assert_array_equal(again["Estimate"].to_numpy(), ref["Estimate"].to_numpy())
    np.testing.assert_array_equal(again["Model_Frequency"].to_numpy(), ref["Model_Frequency"].to_numpy())

    run(eng, "explicit", fast_blocks=[1, 101, 401, 650])
    assert eng._explicit_starts is not None
    again = run(eng, "after_explicit")
    assert eng._explicit_starts is None
    np.testing.assert_array_equal(again["Estimate"].to_numpy(), ref["Estimate"].to_numpy())
    eng.close()

    with pytest.raises(ValueError, match="centered=False"):
        api.device_genotypes(_Dummy(300, 900), centered=False)
    weighted = _Dummy(300, 900)
    weighted._weighted = True
    with pytest.raises(ValueError, match="residual weights"):
        api.device_genotypes(weighted)


class _Dummy:
    """Stand-in with the attributes device_genotypes reads before it touches the device."""
    def __init__(self, n, p):
        self.n, self.p, self.block_size, self._weighted = n, p, 64, False


@pytest.mark.parametrize("method,nranks,groups", [("BayesC", 2, 0), ("BayesR", 3, 0), ("BayesC", 4, 0), ("BayesC", 2, 2), ("BayesR", 2, 4)])
def test_library_sharded_sweep_with_several_ranks_over_the_loopback_transport(method, nranks, groups):
    """jwas_hip_sweep_sharded with MORE THAN ONE rank: marker shards, one context per rank (here engines of one process on
    different host threads; the all-reduce goes through the loopback transport, the same call site RCCL uses).  Per sweep
    every rank sweeps its own markers from the same residual snapshot, packs (fl64(r_local) - fl64(r_snapshot), its marker
    statistics), ONE all-reduce, r = fl32(r_snapshot + sum): the reference's independent-block reconcile (BayesABC.jl:205-253)
    with one block per rank.  Against a single-process emulation on the oracle: the reconciled residual is the same on every
    rank bit for bit, the shards' chains equal the oracle's (indicators exactly), the statistics are the all-rank sums.
    groups > 0: every rank's sweep with grouped launches (jwas_hip_setup_groups on each shard, jwas_sweep_params.group_launch)."""
    import threading
    import jwas_jl_amd as J
    from jwas_jl_amd.dist import shard_range
    bs = 64
    d = make_dataset(n=700, p=64 * 7 + 30, ncausal=8, seed=60 + nranks)
    X, p = d["X"], d["X"].shape[1]
    y = (d["y"] - d["y"].mean()).astype(np.float32)
    if method == "BayesR":
        kw = dict(vare=np.float32(0.5), var_effect=np.float32(0.05), pi_classes=np.array([0.9, 0.05, 0.03, 0.02]))
    else:
        kw = dict(vare=np.float32(0.5), var_effect=np.float32(0.004), pi=0.9)
    shards = [shard_range(p, r, nranks, bs) for r in range(nranks)]
    nsweeps = 5
    # ---- emulation on the oracle: every shard from the same snapshot, then the reconcile
    orcs = []
    for lo, hi in shards:
        o = OracleEngine("lookahead")
        o.load_dense(np.asfortranarray(X[:, lo:hi])); o.setup_blocks(bs); o.init_state(method, 1)
        if groups:
            o.setup_groups(groups)
        if method == "BayesR":
            o.set_state(0, delta=np.ones(hi - lo, dtype=np.int32))
        orcs.append(o)
    r = y.copy()
    ref_stats = []
    for it in range(1, nsweeps + 1):
        snap = r.copy()
        tot = np.zeros(len(r), dtype=np.float64)
        nev, sd = 0.0, 0.0
        for o, (lo, hi) in zip(orcs, shards):
            o.set_residual(snap)
            st = o.sweep(iteration=it, seed=31, marker_offset=lo, group_launch=bool(groups), **kw)
            tot += o.get_residual().astype(np.float64) - snap.astype(np.float64)
            nev += st["n_events"]; sd += st["sum_delta"][0] if method != "BayesR" else st["class_counts"][1:].sum()
        r = (snap.astype(np.float64) + tot).astype(np.float32)
        ref_stats.append((nev, sd, float(r.astype(np.float64) @ r.astype(np.float64))))
    # ---- the library, one thread per rank
    out, errs = [None] * nranks, []

    def run(rank):
        try:
            lo, hi = shards[rank]
            e = J.HipEngine(0)
            e.load_dense(np.asfortranarray(X[:, lo:hi]))
            e.comm_init_loopback(2, rank, nranks)
            assert e.comm_info() == (rank, nranks)
            e.setup_blocks(bs, "f64"); e.init_state(method, 1)
            if groups:
                e.setup_groups(groups, "f64")
            if method == "BayesR":
                e.set_state(0, delta=np.ones(hi - lo, dtype=np.int32))
            e.set_residual(y)
            sts = [e.sweep_sharded(iteration=it, seed=31, marker_offset=lo, group_launch=bool(groups), **kw) for it in range(1, nsweeps + 1)]
            out[rank] = (e.get_state(0), e.get_residual(0), sts)
            e.close()
        except Exception as ex:                          # noqa: BLE001
            errs.append((rank, repr(ex)))

    th = [threading.Thread(target=run, args=(rk,)) for rk in range(nranks)]
    for x in th:
        x.start()
    for x in th:
        x.join(timeout=300)
    assert not errs, errs
    assert all(o is not None for o in out)
    for rk in range(nranks):
        assert np.array_equal(out[rk][1], out[0][1])                                  # the same reconciled residual everywhere
        ao, bo, do = orcs[rk].get_state(0)
        a, b_, dl = out[rk][0]
        assert np.array_equal(dl, do)
        np.testing.assert_allclose(a, ao, rtol=0, atol=5e-6)
        for it in range(nsweeps):
            st = out[rk][2][it]
            assert st["n_events"] == ref_stats[it][0]                                 # all-rank sums on every rank
            got_sd = st["sum_delta"][0] if method != "BayesR" else st["class_counts"][1:].sum()
            assert got_sd == ref_stats[it][1]
            assert st["resid_ss"][0, 0] == pytest.approx(ref_stats[it][2], rel=2e-5)
    np.testing.assert_allclose(out[0][1], r, rtol=0, atol=1e-4)
