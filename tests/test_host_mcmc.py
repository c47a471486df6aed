"""Host logic of the chain (get_genotypes / build_model / runMCMC) on CPU, driven through the
engine-injection point with the oracle engine (tests/oracle_engine.py)."""
import os

import numpy as np
import pandas as pd
import pytest

from conftest import make_dataset
from oracle_engine import OracleEngine
from jwas_jl_amd import api
from jwas_jl_amd.mcmc import genetic2marker


def _six_animals(with_missing=False):
    """The 6 x 4 table of test/unit/test_streaming_codec.jl:6-16."""
    rows = [[0, 1, 2, 0], [1, 0, 1, 2], [2, 9 if with_missing else 1, 0, 1], [0, 2, 1, 0], [1, 1, 2, 2], [2, 0, 0, 1]]
    ids = ["a1", "a2", "a3", "a4", "a5", "a6"]
    df = pd.DataFrame(rows, columns=["m1", "m2", "m3", "m4"], dtype=np.float64)
    df.insert(0, "ID", ids)
    ph = pd.DataFrame({"ID": ids, "y1": np.array([1.1, -0.3, 0.8, -0.9, 0.5, -0.1], dtype=np.float32)})
    return df, ph


def test_get_genotypes_qc_center_and_missing():
    df, _ = _six_animals(with_missing=True)
    g = api.get_genotypes(df, 1.0, method="BayesC", quality_control=True, center=True)
    assert g.nObs == 6 and g.nMarkers == 4 and g.markerID == ["m1", "m2", "m3", "m4"]
    assert g.genotypes.dtype == np.float32 and g.genotypes.flags.f_contiguous
    np.testing.assert_allclose(g.genotypes.mean(axis=0), 0, atol=1e-6)
    # missing (9) -> column mean of the non-missing, i.e. 0 after centring (readgenotypes.jl:372-384)
    assert abs(g.genotypes[2, 1]) < 1e-6
    raw = np.array([1, 0, 2, 1, 0], dtype=np.float64)
    np.testing.assert_allclose(g.alleleFreq[1], raw.mean() / 2, atol=1e-6)
    assert g.sum2pq == pytest.approx(float((2 * g.alleleFreq * (1 - g.alleleFreq)).sum()), rel=1e-5)


def test_get_genotypes_maf_filter_and_errors(tmp_path):
    X = np.array([[0, 1, 2], [0, 1, 0], [0, 2, 1], [0, 0, 2]], dtype=np.float64)     # first locus fixed
    g = api.get_genotypes(X, 1.0)
    assert g.nMarkers == 2 and g.markerID == ["2", "3"]
    with pytest.raises(ValueError, match="storage must be"):
        api.get_genotypes(X, 1.0, storage="disk")
    g64 = api.get_genotypes(X, 1.0, double_precision=True)                              # readgenotypes.jl:298: Float64 on request
    assert g64.genotypes.dtype == np.float64 and g.genotypes.dtype == np.float32
    np.testing.assert_allclose(g64.genotypes, g.genotypes, atol=1e-6)
    with pytest.raises(NotImplementedError, match="GBLUP"):
        api.get_genotypes(X, 1.0, method="GBLUP")
    with pytest.raises(ValueError, match="starting values"):
        api.get_genotypes(X, 1.0, quality_control=False, starting_value=np.zeros(5))
    f = tmp_path / "g.csv"
    f.write_text("ID,m1,m2\na,0,1\nb,1,2\nc,2,0\n")
    gf = api.get_genotypes(str(f), 1.0, quality_control=False)
    assert gf.obsID == ["a", "b", "c"] and gf.markerID == ["m1", "m2"]


def test_genetic2marker_formulas():
    """test/unit/test_annotated_bayesc.jl:452-471: G = Vg / ((1-pi) * sum2pq) (+ BayesR / vector / MT forms)."""
    df = pd.DataFrame({"ID": ["a1", "a2", "a3"], "m1": [0.0, 1.0, 2.0], "m2": [1.0, 1.0, 0.0], "m3": [2.0, 1.0, 1.0]})
    g = api.get_genotypes(df, 2.5, method="BayesC", Pi=0.3, quality_control=False)
    assert genetic2marker(g, 0.3, "BayesC") == pytest.approx(2.5 / ((1 - 0.3) * g.sum2pq), rel=1e-6)
    pi4 = np.array([0.95, 0.03, 0.015, 0.005])
    assert genetic2marker(g, pi4, "BayesR") == pytest.approx(2.5 / (g.sum2pq * (0.03 * 0.01 + 0.015 * 0.1 + 0.005)), rel=1e-6)
    piv = np.array([0.2, 0.5, 0.9])
    af = g.alleleFreq.astype(np.float64)
    assert genetic2marker(g, piv, "BayesC") == pytest.approx(2.5 / float((2 * af * (1 - af) * (1 - piv)).sum()), rel=1e-6)
    with pytest.raises(ValueError, match="length 4"):
        genetic2marker(g, np.array([0.5, 0.5]), "BayesR")
    with pytest.raises(ValueError, match="must have length 3"):
        genetic2marker(g, np.array([0.5, 0.5]), "BayesC")


def _run(form, ph, geno_df, tmp, **kw):
    geno = api.get_genotypes(geno_df, 1.0, method=kw.pop("method", "BayesC"), quality_control=False, center=True,
                             Pi=kw.pop("Pi", 0.0), estimatePi=kw.pop("estimatePi", True))
    model = api.build_model("y1 = intercept + geno", 1.0)
    return api.runMCMC(model, ph, chain_length=40, burnin=10, output_samples_frequency=10, seed=2026,
                       output_folder=os.path.join(tmp, f"res_{form}"), outputEBV=False,
                       _engine=OracleEngine(form), block_size=64, **kw)


def test_runmcmc_dense_vs_block_same_seed(tmp_path):
    """Shape of test/unit/test_streaming_codec.jl:53-105 ("same draws, different storage"): two
    algebraically identical device forms agree on posterior means to 1e-4 on the 6-animal table."""
    geno_df, ph = _six_animals()
    out_d = _run("dense", ph, geno_df, str(tmp_path))
    out_b = _run("block", ph, geno_df, str(tmp_path))
    ed = out_d["marker effects geno"]["Estimate"].to_numpy(dtype=np.float64)
    eb = out_b["marker effects geno"]["Estimate"].to_numpy(dtype=np.float64)
    assert len(ed) == 4
    np.testing.assert_allclose(eb, ed, atol=1e-4)
    assert float(out_b["residual variance"]["Estimate"][0]) == pytest.approx(float(out_d["residual variance"]["Estimate"][0]), abs=1e-4)


def test_runmcmc_same_seed_is_deterministic(tmp_path):
    """test/runtests.jl:302-320: same seed => identical residual-variance estimate."""
    geno_df, ph = _six_animals()
    a = _run("block", ph, geno_df, str(tmp_path / "a"))
    b = _run("block", ph, geno_df, str(tmp_path / "b"))
    assert a["residual variance"]["Estimate"][0] == b["residual variance"]["Estimate"][0]
    assert np.array_equal(a["marker effects geno"]["Estimate"], b["marker effects geno"]["Estimate"])


@pytest.mark.parametrize("method,Pi", [("BayesC", 0.95), ("BayesB", 0.95), ("BayesA", 0.0), ("BayesR", 0.0)])
def test_runmcmc_recovers_signal_and_writes_outputs(tmp_path, method, Pi):
    d = make_dataset(n=300, p=400, ncausal=5, seed=21, center=False)
    ids = [f"id{i}" for i in range(300)]
    gdf = pd.DataFrame(d["raw"], columns=[f"snp{j}" for j in range(400)])
    gdf.insert(0, "ID", ids)
    ph = pd.DataFrame({"ID": ids[::-1], "y1": d["y"][::-1]})                 # phenotype order differs from genotype order
    geno = api.get_genotypes(gdf, method=method, Pi=Pi)
    model = api.build_model("y1 = intercept + geno")
    folder = str(tmp_path / "out")
    out = api.runMCMC(model, ph, chain_length=150, burnin=30, seed=7, output_folder=folder,
                      _engine=OracleEngine("block"), block_size=64)
    me = out["marker effects geno"]
    assert list(me.columns) == ["Trait", "Marker_ID", "Estimate", "SD", "Model_Frequency"]
    assert len(me) == geno.nMarkers
    causal_ids = {f"snp{j}" for j in d["causal"]}
    top = set(me.reindex(me["Estimate"].abs().sort_values(ascending=False).index)["Marker_ID"].head(8))
    assert len(top & causal_ids) >= 3
    ebv = out["EBV_y1"]
    # EBVs are reported for mme.output_ID = all genotyped individuals, in genotype order (input_data_validation.jl:150-154)
    assert list(ebv.columns) == ["ID", "EBV", "PEV"] and list(ebv["ID"]) == ids
    assert np.corrcoef(ebv["EBV"], d["y"])[0, 1] > 0.5
    assert 0 < float(out["residual variance"]["Estimate"][0]) < 2 * float(np.var(d["y"]))
    for f in ("marker_effects_geno.txt", "residual_variance.txt", "location_parameters.txt", "EBV_y1.txt",
              "MCMC_samples_residual_variance.txt", "MCMC_samples_marker_effects_geno_y1.txt",
              "IDs_for_individuals_with_genotypes.txt", "IDs_for_individuals_with_phenotypes.txt"):
        assert os.path.exists(os.path.join(folder, f)), f
    if method in ("BayesC", "BayesB", "BayesR"):
        assert "pi_geno" in out
    # an existing folder is never overwritten (JWAS.jl:255-262)
    geno2 = api.get_genotypes(gdf, method=method, Pi=Pi)
    model2 = api.build_model("y1 = intercept + geno2", genotypes={"geno2": geno2})
    api.runMCMC(model2, ph, chain_length=2, seed=7, output_folder=folder, _engine=OracleEngine("block"), block_size=64)
    assert os.path.isdir(folder + "1")


def test_runmcmc_multitrait_and_fixed_effects(tmp_path):
    d = make_dataset(n=250, p=192, ncausal=4, seed=5, center=False)
    rng = np.random.default_rng(1)
    ids = [str(i) for i in range(250)]
    x1 = rng.standard_normal(250)
    sex = np.where(rng.random(250) < 0.5, "m", "f")
    y1 = d["y"] + 0.8 * x1 + np.where(sex == "m", 0.5, 0.0)
    y2 = 0.6 * d["y"] + rng.standard_normal(250) * 0.5
    ph = pd.DataFrame({"ID": ids, "y1": y1.astype(np.float32), "y2": y2.astype(np.float32), "x1": x1, "sex": sex})
    gdf = pd.DataFrame(d["raw"], columns=[f"s{j}" for j in range(192)])
    gdf.insert(0, "ID", ids)
    geno = api.get_genotypes(gdf, method="BayesC", Pi=0.0, estimatePi=True)
    model = api.build_model("y1 = intercept + x1 + sex + geno\ny2 = intercept + geno")
    api.set_covariate(model, "x1")
    out = api.runMCMC(model, ph, chain_length=60, burnin=10, seed=3, output_folder=str(tmp_path / "mt"),
                      _engine=OracleEngine("block"), block_size=64)
    lp = out["location parameters"]
    assert set(lp["Trait"]) == {"y1", "y2"} and {"intercept", "x1", "sex"} <= set(lp["Effect"])
    assert float(lp[(lp.Trait == "y1") & (lp.Effect == "x1")]["Estimate"].iloc[0]) == pytest.approx(0.8, abs=0.25)
    assert len(out["marker effects geno"]) == 2 * geno.nMarkers
    assert out["residual variance"].shape[0] == 4 and out["pi_geno"].shape[0] == 4
    assert "EBV_y1" in out and "EBV_y2" in out


def _two_trait(n=250, p=192, seed=5):
    d = make_dataset(n=n, p=p, ncausal=4, seed=seed, center=False)
    rng = np.random.default_rng(1)
    ids = [str(i) for i in range(n)]
    y2 = 0.6 * d["y"] + rng.standard_normal(n) * 0.5
    ph = pd.DataFrame({"ID": ids, "y1": d["y"].astype(np.float32), "y2": y2.astype(np.float32)})
    gdf = pd.DataFrame(d["raw"], columns=[f"s{j}" for j in range(p)])
    gdf.insert(0, "ID", ids)
    return d, ph, gdf


def test_runmcmc_multitrait_sampler_II(tmp_path):
    """multi_trait_sampler=:II (readgenotypes.jl:227, MTBayesABC.jl:129-210): joint-state sampler; with the
    restrictive all-or-none prior only states 00 and 11 keep mass."""
    d, ph, gdf = _two_trait()
    Pi = {(0.0, 0.0): 0.7, (1.0, 0.0): 0.0, (0.0, 1.0): 0.0, (1.0, 1.0): 0.3}
    geno = api.get_genotypes(gdf, method="BayesC", Pi=Pi, estimatePi=False, multi_trait_sampler="II")
    model = api.build_model("y1 = intercept + geno\ny2 = intercept + geno")
    out = api.runMCMC(model, ph, chain_length=60, burnin=10, seed=3, output_folder=str(tmp_path / "mt2"),
                      _engine=OracleEngine("block"), block_size=64)
    me = out["marker effects geno"]
    f1 = me[me.Trait == "y1"]["Model_Frequency"].to_numpy()
    f2 = me[me.Trait == "y2"]["Model_Frequency"].to_numpy()
    assert np.array_equal(f1, f2)                       # a locus affects both traits or none
    assert 0 < f1.mean() < 1
    causal = [f"s{j}" for j in d["causal"]]
    top = set(me[me.Trait == "y1"].sort_values("Model_Frequency", ascending=False)["Marker_ID"].head(10))
    assert len(top & set(causal)) >= 2


def test_runmcmc_constraint_true_runs_mega_path(tmp_path):
    """constraint=true on G and R (readgenotypes.jl:219, build_MME.jl): megaBayesABC! = independent single-trait
    chains, diagonal variance draws, one pi per trait (MCMC_BayesianAlphabet.jl:233-234,300-301)."""
    d, ph, gdf = _two_trait()
    geno = api.get_genotypes(gdf, method="BayesC", Pi={(0.0, 0.0): 0.8, (1.0, 0.0): 0.05, (0.0, 1.0): 0.05, (1.0, 1.0): 0.1},
                             estimatePi=True, constraint=True)
    model = api.build_model("y1 = intercept + geno\ny2 = intercept + geno", constraint=True)
    out = api.runMCMC(model, ph, chain_length=60, burnin=10, seed=3, output_folder=str(tmp_path / "mega"),
                      _engine=OracleEngine("block"), block_size=64)
    rv = out["residual variance"]["Estimate"].to_numpy().reshape(2, 2)
    gv = out["marker effects variance geno"]["Estimate"].to_numpy().reshape(2, 2)
    assert rv[0, 1] == 0 and rv[1, 0] == 0 and gv[0, 1] == 0 and gv[1, 0] == 0
    assert rv[0, 0] > 0 and gv[1, 1] > 0
    assert list(out["pi_geno"]["π"]) == ["y1", "y2"]
    assert np.all((out["pi_geno"]["Estimate"] > 0.5) & (out["pi_geno"]["Estimate"] < 1))
    single = api.get_genotypes(gdf, method="BayesC", Pi=0.5, constraint=True)
    m1 = api.build_model("y1 = intercept + single", genotypes={"single": single})
    with pytest.raises(ValueError, match="constraint==true is for multi-trait only"):
        api.runMCMC(m1, ph, chain_length=2, output_folder=str(tmp_path / "e"), _engine=OracleEngine("block"), block_size=64)


def test_runmcmc_rrblup_constraint_true_runs_mega_path(tmp_path):
    """Multi-trait RR-BLUP with constraint = true: megaBayesC0! (MCMC_BayesianAlphabet.jl:260-262) = the constrained chains
    with every marker in the model for every trait (pi_k = 0), diagonal variance draws."""
    d, ph, gdf = _two_trait()
    geno = api.get_genotypes(gdf, method="RR-BLUP", constraint=True)
    model = api.build_model("y1 = intercept + geno\ny2 = intercept + geno", constraint=True)
    used = {}

    class Spy(OracleEngine):
        def init_state(self, m, t=1):
            used["method"] = m
            return super().init_state(m, t)

        def sweep(self, **kw):
            used["pi"] = np.asarray(kw["pi"]).copy()
            return super().sweep(**kw)

    out = api.runMCMC(model, ph, chain_length=40, burnin=10, seed=3, output_folder=str(tmp_path / "rr"), _engine=Spy("block"), block_size=64)
    assert used["method"] == "MegaBayesC" and np.array_equal(used["pi"], [0.0, 0.0])
    gv = out["marker effects variance geno"]["Estimate"].to_numpy().reshape(2, 2)
    assert gv[0, 1] == 0 and gv[1, 0] == 0 and gv[0, 0] > 0 and gv[1, 1] > 0
    assert (out["marker effects geno"]["Model_Frequency"].to_numpy() == 1.0).all()


def test_runmcmc_contract_errors(tmp_path):
    geno_df, ph = _six_animals()
    geno = api.get_genotypes(geno_df, 1.0, quality_control=False)
    model = api.build_model("y1 = intercept + geno", 1.0)
    with pytest.raises(ValueError, match="independent_blocks=true requires fast_blocks"):
        api.runMCMC(model, ph, independent_blocks=True, output_folder=str(tmp_path / "e1"))     # JWAS.jl:242-244
    with pytest.raises(NotImplementedError, match="single_step_analysis"):
        api.runMCMC(model, ph, single_step_analysis=True, output_folder=str(tmp_path / "e2"))
    with pytest.raises(ValueError, match="at least two block starts"):
        api.runMCMC(model, ph, fast_blocks=4, output_folder=str(tmp_path / "e3"), _engine=OracleEngine("block"))   # 4 markers: one start
    with pytest.raises(ValueError, match="Model equations are wrong"):
        api.build_model("")
    nog = api.build_model("y1 = intercept")
    with pytest.raises(NotImplementedError, match="genotype term"):
        api.runMCMC(nog, ph, output_folder=str(tmp_path / "e4"))


def test_fast_blocks_rescales_chain_length(tmp_path):
    """JWAS.jl:293-316: a numeric fast_blocks divides chain_length by the block size and runs
    block-size within-block repetitions; burnin is not rescaled."""
    d = make_dataset(n=200, p=200, ncausal=3, seed=9, center=False)
    ids = [str(i) for i in range(200)]
    gdf = pd.DataFrame(d["raw"]); gdf.insert(0, "ID", ids)
    ph = pd.DataFrame({"ID": ids, "y1": d["y"]})
    geno = api.get_genotypes(gdf, method="BayesC", Pi=0.9)
    model = api.build_model("y1 = intercept + geno")
    out = api.runMCMC(model, ph, chain_length=640, burnin=2, fast_blocks=64, seed=1, outputEBV=False,
                      output_folder=str(tmp_path / "fb"), _engine=OracleEngine("block"))
    assert out["_timing"]["iterations"] == 10 and out["_timing"]["block_size"] == 64


def test_independent_blocks_through_runmcmc(tmp_path):
    """runMCMC(fast_blocks=..., independent_blocks=true) (JWAS.jl:242-244, MCMC_BayesianAlphabet.jl:251)."""
    d = make_dataset(n=200, p=260, ncausal=3, seed=9, center=False)
    ids = [str(i) for i in range(200)]
    gdf = pd.DataFrame(d["raw"]); gdf.insert(0, "ID", ids)
    ph = pd.DataFrame({"ID": ids, "y1": d["y"]})
    geno = api.get_genotypes(gdf, method="BayesC", Pi=0.9)
    model = api.build_model("y1 = intercept + geno")
    out = api.runMCMC(model, ph, chain_length=64 * 12, burnin=2, fast_blocks=64, independent_blocks=True, seed=1,
                      output_folder=str(tmp_path / "ib"), _engine=OracleEngine("block"))
    assert out["_timing"]["iterations"] == 12
    assert np.corrcoef(out["EBV_y1"]["EBV"], ph["y1"])[0, 1] > 0.4


def test_adaptive_block_size_policy(tmp_path):
    """Sparse priors keep block sizes 512 and 1024 resident; the next sweep's size follows the last sweep's number of
    effect changes (mcmc.pick_block_size) -- a chain quantity, so the run is reproducible."""
    from jwas_jl_amd.mcmc import pick_block_size
    assert pick_block_size(100, 600_000) == 1024 and pick_block_size(20_000, 600_000) == 512

    class Spy(OracleEngine):
        def __init__(self):
            super().__init__("block")
            self.sizes = []

        def sweep(self, **kw):
            self.sizes.append(self.block_size)
            return super().sweep(**kw)

    d = make_dataset(n=120, p=4300, ncausal=4, seed=3, center=False)
    ids = [str(i) for i in range(120)]
    gdf = pd.DataFrame(d["raw"]); gdf.insert(0, "ID", ids)
    ph = pd.DataFrame({"ID": ids, "y1": d["y"]})
    outs = []
    for rep in range(2):
        geno = api.get_genotypes(gdf, method="BayesC", Pi=0.99)
        model = api.build_model("y1 = intercept + geno")
        spy = Spy()
        outs.append(api.runMCMC(model, ph, chain_length=12, burnin=2, seed=5, outputEBV=False,
                                output_folder=str(tmp_path / f"ad{rep}"), _engine=spy))
        assert spy.sizes[0] == 512 and set(spy.sizes) <= {512, 1024} and 1024 in spy.sizes
    assert np.array_equal(outs[0]["marker effects geno"]["Estimate"], outs[1]["marker effects geno"]["Estimate"])


def test_heterogeneous_residuals_weights(tmp_path):
    """runMCMC(heterogeneous_residuals=true): invweights = 1 ./ df.weights (build_MME.jl:305-310) enters the marker
    sweep, the location-parameter equations and the residual-variance draw."""
    d = make_dataset(n=220, p=150, ncausal=4, seed=14, center=False)
    rng = np.random.default_rng(2)
    ids = [str(i) for i in range(220)]
    wts = rng.uniform(0.5, 4.0, 220)
    y = d["y"] + rng.standard_normal(220) * np.sqrt(wts) * 0.3              # noisier records carry larger weights
    gdf = pd.DataFrame(d["raw"]); gdf.insert(0, "ID", ids)
    ph = pd.DataFrame({"ID": ids, "y1": y.astype(np.float32), "weights": wts})
    outs = {}
    for het in (False, True):
        geno = api.get_genotypes(gdf, method="BayesC", Pi=0.9)
        model = api.build_model("y1 = intercept + geno")
        outs[het] = api.runMCMC(model, ph, chain_length=40, burnin=5, seed=9, heterogeneous_residuals=het,
                                output_folder=str(tmp_path / f"w{het}"), _engine=OracleEngine("block"), block_size=64)
    a, b = outs[False]["marker effects geno"]["Estimate"], outs[True]["marker effects geno"]["Estimate"]
    assert not np.allclose(a, b)
    assert np.corrcoef(outs[True]["EBV_y1"]["EBV"], ph["y1"])[0, 1] > 0.4
    with pytest.raises(ValueError, match="requires a column named weights"):
        geno = api.get_genotypes(gdf, method="BayesC", Pi=0.9)
        model = api.build_model("y1 = intercept + geno")
        api.runMCMC(model, ph.drop(columns="weights"), chain_length=2, heterogeneous_residuals=True,
                    output_folder=str(tmp_path / "w_err"), _engine=OracleEngine("block"), block_size=64)


def test_config1_plumbing_on_the_oracle(tmp_path, config1_data):
    """BASELINE.json configs[0]: single-trait BayesC pi=0.95, 500 x 2000, 1000 iterations, CPU only (the oracle engine
    drives the same host loop the device path uses): the chain runs, outputs have the reference's shape, QTL are found."""
    d = config1_data
    n, p = d["X"].shape
    ids = [f"i{i}" for i in range(n)]
    gdf = pd.DataFrame(d["raw"], columns=[f"m{j}" for j in range(p)])
    gdf.insert(0, "ID", ids)
    ph = pd.DataFrame({"ID": ids, "y1": d["y"]})
    geno = api.get_genotypes(gdf, method="BayesC", Pi=0.95, estimatePi=True)
    model = api.build_model("y1 = intercept + geno")
    out = api.runMCMC(model, ph, chain_length=1000, burnin=100, seed=2026, output_folder=str(tmp_path / "c1"),
                      _engine=OracleEngine("block"), block_size=256)
    me = out["marker effects geno"]
    assert len(me) == geno.nMarkers and out["_timing"]["iterations"] == 1000
    top = set(me.reindex(me["Model_Frequency"].sort_values(ascending=False).index)["Marker_ID"].head(20))
    assert len(top & {f"m{j}" for j in d["causal"]}) >= 8
    assert 0.9 < float(out["pi_geno"]["Estimate"][0]) < 1.0
    h2 = 1 - float(out["residual variance"]["Estimate"][0]) / float(np.var(d["y"]))
    assert 0.25 < h2 < 0.75
    assert np.corrcoef(out["EBV_y1"]["EBV"], ph["y1"])[0, 1] > 0.6


def test_ebv_for_individuals_without_records_and_outputEBV_list(tmp_path):
    """Held-out prediction, the reference's standard use: individuals with genotypes but no phenotype get EBVs
    (default output_ID = all genotyped individuals, input_data_validation.jl:150-154; Mi.output_genotypes,
    tools4genotypes.jl:290-296); outputEBV(model, IDs) restricts the list (output.jl:60-70) and drops IDs without
    genotypes (input_data_validation.jl:181-186)."""
    d = make_dataset(n=300, p=200, ncausal=8, seed=77, center=False)
    ids = [f"id{i}" for i in range(300)]
    gdf = pd.DataFrame(d["raw"], columns=[f"snp{j}" for j in range(200)])
    gdf.insert(0, "ID", ids)
    y = d["y"].astype(np.float64).copy()
    held = np.arange(0, 300, 5)
    y_masked = y.copy()
    y_masked[held] = np.nan
    ph = pd.DataFrame({"ID": ids, "y1": y_masked})
    geno = api.get_genotypes(gdf, method="BayesC", Pi=0.9)
    model = api.build_model("y1 = intercept + geno")
    out = api.runMCMC(model, ph, chain_length=200, burnin=40, seed=3, output_folder=str(tmp_path / "a"),
                      _engine=OracleEngine("block"), block_size=64)
    ebv = out["EBV_y1"]
    assert list(ebv["ID"]) == ids                                   # all genotyped individuals, genotype order
    assert np.corrcoef(ebv["EBV"].to_numpy()[held], y[held])[0, 1] > 0.3          # held-out prediction
    # EBV of a training individual = its row of X times the posterior mean effects (linearity)
    me = out["marker effects geno"]["Estimate"].to_numpy(dtype=np.float64)
    np.testing.assert_allclose(ebv["EBV"].to_numpy(), np.asarray(geno.genotypes, dtype=np.float64) @ me, atol=2e-3)

    geno = api.get_genotypes(gdf, method="BayesC", Pi=0.9)
    model = api.build_model("y1 = intercept + geno")
    api.outputEBV(model, ["id5", "id0", "nobody", "id7"])
    out2 = api.runMCMC(model, ph, chain_length=200, burnin=40, seed=3, output_folder=str(tmp_path / "b"),
                       _engine=OracleEngine("block"), block_size=64)
    assert list(out2["EBV_y1"]["ID"]) == ["id5", "id0", "id7"]
    full = ebv.set_index("ID")["EBV"]
    np.testing.assert_allclose(out2["EBV_y1"]["EBV"].to_numpy(), full.loc[["id5", "id0", "id7"]].to_numpy(), atol=1e-5)


def test_impute_missing_residuals_conditional_moments():
    """sampleMissingResiduals (residual.jl:52-73): E[e_m | e_o] = Rc Ro^-1 e_o, Var = Rmm - Rc Ro^-1 Rc';
    the per-record weights are inv(R0[o,o]) embedded in zeros (getRi, residual.jl:2-11)."""
    from jwas_jl_amd.mcmc import _impute_missing_residuals
    rng = np.random.default_rng(0)
    R0 = np.array([[2.0, 0.8, 0.3], [0.8, 1.5, -0.4], [0.3, -0.4, 1.0]])
    n = 60000
    observed = np.ones((n, 3), dtype=bool)
    observed[: n // 2, 1] = False                    # pattern (1,0,1)
    observed[n // 2:, 0] = False
    observed[n // 2:, 2] = False                     # pattern (0,1,0)
    e = [np.full(n, 0.7), np.full(n, -0.2), np.full(n, 1.1)]
    res = [v.copy() for v in e]
    Ri = _impute_missing_residuals(res, observed, R0, rng)
    # observed entries untouched
    assert np.array_equal(res[0][: n // 2], e[0][: n // 2]) and np.array_equal(res[1][n // 2:], e[1][n // 2:])
    o = np.array([True, False, True])
    Ro_inv = np.linalg.inv(R0[np.ix_(o, o)])
    Rc = R0[np.ix_(~o, o)]
    mean = (Rc @ Ro_inv @ np.array([0.7, 1.1]))[0]
    var = (R0[1, 1] - Rc @ Ro_inv @ Rc.T)[0, 0]
    assert abs(res[1][: n // 2].mean() - mean) < 4 * np.sqrt(var / (n // 2))
    assert abs(res[1][: n // 2].var() - var) < 0.03 * var
    # pattern (0,1,0): two missing traits, joint conditional covariance
    o2 = np.array([False, True, False])
    Rc2 = R0[np.ix_(~o2, o2)]
    cov2 = R0[np.ix_(~o2, ~o2)] - Rc2 @ Rc2.T / R0[1, 1]
    got = np.cov(np.stack([res[0][n // 2:], res[2][n // 2:]]))
    np.testing.assert_allclose(got, cov2, atol=0.03)
    want = np.zeros((3, 3)); want[np.ix_(o, o)] = Ro_inv
    np.testing.assert_allclose(Ri[0], want)
    want2 = np.zeros((3, 3)); want2[1, 1] = 1 / R0[1, 1]
    np.testing.assert_allclose(Ri[-1], want2)


def test_multitrait_with_partially_missing_records(tmp_path):
    """The reference's multi-trait tests run on records with missing traits (demo_7animals: a5 has y1 but no y2;
    test_multitrait_mcmc.jl:111-128): such records stay in the analysis, their missing residuals are imputed."""
    d = make_dataset(n=260, p=150, ncausal=6, seed=12, center=False)
    ids = [f"id{i}" for i in range(260)]
    gdf = pd.DataFrame(d["raw"], columns=[f"snp{j}" for j in range(150)])
    gdf.insert(0, "ID", ids)
    rng = np.random.default_rng(5)
    y1 = d["y"].astype(np.float64)
    y2 = 0.6 * y1 + 0.8 * rng.standard_normal(260)
    y1m, y2m = y1.copy(), y2.copy()
    y2m[::3] = np.nan                        # partially missing
    y1m[1::7] = np.nan
    both = np.arange(5, 260, 41)
    y1m[both] = np.nan; y2m[both] = np.nan   # no record at all: dropped from the analysis, still gets an EBV
    ph = pd.DataFrame({"ID": ids, "y1": y1m, "y2": y2m})
    geno = api.get_genotypes(gdf, np.eye(2) * 0.5, method="BayesC", Pi={(0.0, 0.0): 0.7, (1.0, 0.0): 0.1, (0.0, 1.0): 0.1, (1.0, 1.0): 0.1})
    model = api.build_model("y1 = intercept + geno\ny2 = intercept + geno", np.eye(2))
    out = api.runMCMC(model, ph, chain_length=150, burnin=30, seed=3, output_folder=str(tmp_path / "mt"),
                      _engine=OracleEngine("block"), block_size=64)
    assert len(out["residual variance"]) == 4
    assert np.isfinite(out["residual variance"]["Estimate"]).all()
    n_used = len(open(tmp_path / "mt" / "IDs_for_individuals_with_phenotypes.txt").read().split())
    assert n_used == int((~(np.isnan(y1m) & np.isnan(y2m))).sum()) < 260 - len(both) + 1
    ebv = out["EBV_y2"]
    assert list(ebv["ID"]) == ids
    miss2 = np.isnan(y2m)
    assert np.corrcoef(ebv["EBV"].to_numpy()[miss2], y2[miss2])[0, 1] > 0.2      # prediction of the hidden records
    with pytest.raises(ValueError, match="missing_phenotypes=false"):
        geno = api.get_genotypes(gdf, np.eye(2) * 0.5, method="BayesC")
        model = api.build_model("y1 = intercept + geno\ny2 = intercept + geno", np.eye(2))
        api.runMCMC(model, ph, chain_length=5, seed=3, output_folder=str(tmp_path / "mt2"), _engine=OracleEngine("block"),
                    block_size=64, missing_phenotypes=False)


def test_rrblup_is_bayesc_with_all_markers_in_the_model(tmp_path):
    """RR-BLUP runs on the device's BayesC path with pi = 0 fixed (same full conditionals: BayesC0L.jl:20-47 vs
    BayesABC.jl:24-58 with probDelta1 = 1; variance update with nloci = nMarkers, variance_components.jl:160-162)."""
    d = make_dataset(n=200, p=120, ncausal=6, seed=8, center=False)
    ids = [f"id{i}" for i in range(200)]
    gdf = pd.DataFrame(d["raw"], columns=[f"snp{j}" for j in range(120)])
    gdf.insert(0, "ID", ids)
    ph = pd.DataFrame({"ID": ids, "y1": d["y"]})
    outs = {}
    for tag, kw in (("rr", dict(method="RR-BLUP")), ("c0", dict(method="BayesC", Pi=0.0, estimatePi=False))):
        geno = api.get_genotypes(gdf, **kw)
        model = api.build_model("y1 = intercept + geno")
        outs[tag] = api.runMCMC(model, ph, chain_length=60, burnin=10, seed=4, output_folder=str(tmp_path / tag),
                                _engine=OracleEngine("block"), block_size=64)
    np.testing.assert_array_equal(outs["rr"]["marker effects geno"]["Estimate"].to_numpy(),
                                  outs["c0"]["marker effects geno"]["Estimate"].to_numpy())
    assert (outs["rr"]["marker effects geno"]["Model_Frequency"] == 1.0).all()
    assert "pi_geno" not in outs["rr"]


def test_bayes_lasso_runs_on_the_bayesb_device_path(tmp_path):
    """BayesL (BayesC0L.jl:25-47; gamma_j update variance_components.jl:191-203; G/8 and Gamma(1,8) start
    MCMC_BayesianAlphabet.jl:70-81): all markers in the model, common scale reported, signal recovered."""
    d = make_dataset(n=300, p=250, ncausal=6, seed=19, center=False)
    ids = [f"id{i}" for i in range(300)]
    gdf = pd.DataFrame(d["raw"], columns=[f"snp{j}" for j in range(250)])
    gdf.insert(0, "ID", ids)
    ph = pd.DataFrame({"ID": ids, "y1": d["y"]})
    geno = api.get_genotypes(gdf, method="BayesL")
    model = api.build_model("y1 = intercept + geno")
    out = api.runMCMC(model, ph, chain_length=300, burnin=60, seed=2, output_folder=str(tmp_path / "bl"),
                      _engine=OracleEngine("block"), block_size=64)
    me = out["marker effects geno"]
    assert (me["Model_Frequency"] == 1.0).all()
    assert "marker effects variance geno" in out and float(out["marker effects variance geno"]["Estimate"][0]) > 0
    assert "pi_geno" not in out
    causal = {f"snp{j}" for j in d["causal"]}
    top = set(me.reindex(me["Estimate"].abs().sort_values(ascending=False).index)["Marker_ID"].head(10))
    assert len(top & causal) >= 3
    assert np.corrcoef(out["EBV_y1"]["EBV"], d["y"])[0, 1] > 0.6
    assert os.path.exists(tmp_path / "bl" / "MCMC_samples_marker_effects_variances_geno.txt")


def _annotated_dataset(seed=3, n=400, p=300):
    """Causal markers are enriched among the markers carrying a binary annotation."""
    rng = np.random.default_rng(seed)
    f = rng.uniform(0.1, 0.4, p)
    X = (rng.random((n, p)) < f).astype(np.float32) + (rng.random((n, p)) < f).astype(np.float32)
    ann = np.zeros((p, 2))
    ann[:60, 0] = 1.0                                   # informative annotation
    ann[:, 1] = rng.standard_normal(p)                  # noise annotation
    causal = rng.choice(60, 24, replace=False)
    beta = np.zeros(p); beta[causal] = rng.standard_normal(24)
    g = (X - X.mean(0)) @ beta
    y = 1.0 + g / g.std() * np.sqrt(0.6) + rng.standard_normal(n) * np.sqrt(0.4)
    ids = [f"id{i}" for i in range(n)]
    gdf = pd.DataFrame(X, columns=[f"snp{j}" for j in range(p)])
    gdf.insert(0, "ID", ids)
    return gdf, pd.DataFrame({"ID": ids, "y1": y}), ann, causal


@pytest.mark.parametrize("method", ["BayesC", "BayesR"])
def test_annotated_single_trait_priors(tmp_path, method):
    """Annotated BayesC / BayesR (annotation_updates.jl, annotation_setup.jl): the probit regression of the inclusion
    indicators on the annotations finds the informative annotation; marker-level priors reach the device sweep
    (pi_vec / p x 4 pi_matrix)."""
    gdf, ph, ann, causal = _annotated_dataset()
    kw = dict(Pi=0.9) if method == "BayesC" else dict(Pi=[0.9, 0.06, 0.03, 0.01])
    geno = api.get_genotypes(gdf, method=method, annotations=ann, estimatePi=False, **kw)
    assert geno.estimatePi is True                       # forced (readgenotypes.jl:152-158)
    model = api.build_model("y1 = intercept + geno")
    out = api.runMCMC(model, ph, chain_length=400, burnin=100, seed=7, output_folder=str(tmp_path / method),
                      _engine=OracleEngine("block"), block_size=64)
    tab = out["annotation coefficients geno"]
    if method == "BayesC":
        assert list(tab.columns) == ["Annotation", "Estimate", "SD"]
        assert list(tab["Annotation"]) == ["Intercept", "Annotation_1", "Annotation_2"]
        slope = float(tab["Estimate"][1])
        pi_tab = out["pi_geno"]
        assert len(pi_tab) == geno.nMarkers
        assert pi_tab["Estimate"][:60].mean() < pi_tab["Estimate"][60:].mean() - 0.02     # annotated markers: lower Pr(zero)
    else:
        assert list(tab.columns) == ["Annotation", "Step", "Estimate", "SD"] and len(tab) == 9
        slope = float(tab[(tab["Annotation"] == "Annotation_1") & (tab["Step"] == "step1_zero_vs_nonzero")]["Estimate"].iloc[0])
        assert len(out["pi_geno"]) == 4
    assert slope > 0.3, tab
    me = out["marker effects geno"]
    assert me["Model_Frequency"][:60].mean() > me["Model_Frequency"][60:].mean()


def test_annotation_input_errors():
    """readgenotypes.jl:56-105"""
    gdf, ph, ann, _ = _annotated_dataset(p=80, n=50)
    with pytest.raises(ValueError, match="only supported with method"):
        api.get_genotypes(gdf, method="BayesB", annotations=ann)
    with pytest.raises(ValueError, match="must match the number of raw markers"):
        api.get_genotypes(gdf, method="BayesC", annotations=ann[:-1])
    bad = ann.copy(); bad[:, 0] = 2.0
    with pytest.raises(ValueError, match=r"constant column\(s\) \[1\]"):
        api.get_genotypes(gdf, method="BayesC", annotations=bad)
    col = np.hstack([ann, ann[:, :1] * 2])
    with pytest.raises(ValueError, match="collinear"):
        api.get_genotypes(gdf, method="BayesC", annotations=col)
    with pytest.raises(ValueError, match="positive prior mass in classes 3 or 4"):
        api.get_genotypes(gdf, method="BayesR", annotations=ann, Pi=[0.9, 0.1, 0.0, 0.0])
    with pytest.raises(ValueError, match="Pi vector length"):
        api.get_genotypes(gdf, method="BayesC", annotations=ann, Pi=np.full(7, 0.9))


def test_annotated_two_trait_bayesc(tmp_path):
    """Annotated 2-trait BayesC (annotation_setup.jl:101-133, annotation_updates.jl:287-326,353-361): the tree of probit
    models over the joint states 00/10/01/11 drives a marker-specific joint prior (device: log_prior_states_matrix)."""
    gdf, ph, ann, causal = _annotated_dataset(seed=11, n=350, p=260)
    rng = np.random.default_rng(2)
    y1 = ph["y1"].to_numpy()
    ph = ph.assign(y2=0.7 * y1 + 0.7 * rng.standard_normal(len(y1)))
    Pi = {(0.0, 0.0): 0.85, (1.0, 0.0): 0.05, (0.0, 1.0): 0.05, (1.0, 1.0): 0.05}
    geno = api.get_genotypes(gdf, np.eye(2) * 0.5, method="BayesC", annotations=ann, Pi=Pi)
    model = api.build_model("y1 = intercept + geno\ny2 = intercept + geno", np.eye(2))
    out = api.runMCMC(model, ph, chain_length=300, burnin=80, seed=5, output_folder=str(tmp_path / "amt"),
                      _engine=OracleEngine("block"), block_size=64)
    tab = out["annotation coefficients geno"]
    assert list(tab.columns) == ["Annotation", "Step", "Estimate", "SD"] and len(tab) == 9
    assert set(tab["Step"]) == {"step1_zero_vs_active", "step2_11_vs_singleton", "step3_10_vs_01"}
    slope = float(tab[(tab["Annotation"] == "Annotation_1") & (tab["Step"] == "step1_zero_vs_active")]["Estimate"].iloc[0])
    assert slope > 0.3, tab
    assert len(out["pi_geno"]) == 4 and abs(out["pi_geno"]["Estimate"].sum() - 1.0) < 1e-6
    me = out["marker effects geno"]
    mf = me[me["Trait"] == "y1"]["Model_Frequency"].to_numpy()
    assert mf[:60].mean() > mf[60:].mean()
    with pytest.raises(ValueError, match="shared state 11"):
        geno = api.get_genotypes(gdf, np.eye(2) * 0.5, method="BayesC", annotations=ann,
                                 Pi={(0.0, 0.0): 0.9, (1.0, 0.0): 0.05, (0.0, 1.0): 0.05, (1.0, 1.0): 0.0})
        model = api.build_model("y1 = intercept + geno\ny2 = intercept + geno", np.eye(2))
        api.runMCMC(model, ph, chain_length=5, seed=5, output_folder=str(tmp_path / "amt2"), _engine=OracleEngine("block"), block_size=64)


def test_annotated_two_trait_bayesc_default_pi(tmp_path):
    """Pi omitted (the reference's supported default, annotation_setup.jl:109-118): every marker starts in the all-active
    state 11; a non-zero scalar Pi is the reference's error."""
    gdf, ph, ann, causal = _annotated_dataset(seed=12, n=200, p=130)
    rng = np.random.default_rng(3)
    ph = ph.assign(y2=0.7 * ph["y1"].to_numpy() + 0.7 * rng.standard_normal(len(ph)))
    geno = api.get_genotypes(gdf, np.eye(2) * 0.5, method="BayesC", annotations=ann)
    model = api.build_model("y1 = intercept + geno\ny2 = intercept + geno", np.eye(2))
    out = api.runMCMC(model, ph, chain_length=40, burnin=10, seed=5, output_folder=str(tmp_path / "d"),
                      _engine=OracleEngine("block"), block_size=64)
    assert len(out["pi_geno"]) == 4 and abs(out["pi_geno"]["Estimate"].sum() - 1.0) < 1e-6
    assert np.isfinite(out["marker effects geno"]["Estimate"]).all()
    geno = api.get_genotypes(gdf, np.eye(2) * 0.5, method="BayesC", annotations=ann, Pi=0.5)
    model = api.build_model("y1 = intercept + geno\ny2 = intercept + geno", np.eye(2))
    with pytest.raises(ValueError, match="requires Pi=0.0 or a joint Pi dictionary"):
        api.runMCMC(model, ph, chain_length=5, seed=5, output_folder=str(tmp_path / "e"), _engine=OracleEngine("block"), block_size=64)


def test_heritability_output_matches_its_definition(tmp_path):
    """output.jl:498-512: per saved sample genetic variance = var(EBV) over the output individuals and
    h2 = genVar / (genVar + vare); the tables are the mean / std of those samples (output.jl:196-209)."""
    d = make_dataset(n=150, p=90, ncausal=5, seed=2, center=False)
    ids = [f"id{i}" for i in range(150)]
    gdf = pd.DataFrame(d["raw"], columns=[f"snp{j}" for j in range(90)]); gdf.insert(0, "ID", ids)
    ph = pd.DataFrame({"ID": ids, "y1": d["y"]})
    geno = api.get_genotypes(gdf, method="BayesC", Pi=0.8)
    model = api.build_model("y1 = intercept + geno")
    folder = tmp_path / "h"
    out = api.runMCMC(model, ph, chain_length=60, burnin=10, output_samples_frequency=5, seed=1, output_folder=str(folder),
                      _engine=OracleEngine("block"), block_size=64)
    a = pd.read_csv(folder / "MCMC_samples_marker_effects_geno_y1.txt").to_numpy()
    ve = pd.read_csv(folder / "MCMC_samples_residual_variance.txt").to_numpy().ravel()
    X = np.asarray(geno.genotypes, dtype=np.float64)
    gv = np.array([np.var(X @ a[i], ddof=1) for i in range(len(a))])
    np.testing.assert_allclose(pd.read_csv(folder / "MCMC_samples_genetic_variance.txt").to_numpy().ravel(), gv, rtol=1e-4)
    np.testing.assert_allclose(pd.read_csv(folder / "MCMC_samples_heritability.txt").to_numpy().ravel(), gv / (gv + ve), rtol=1e-4)
    assert float(out["heritability"]["Estimate"][0]) == pytest.approx(float((gv / (gv + ve)).mean()), rel=1e-4)
    geno = api.get_genotypes(gdf, method="BayesC", Pi=0.8)
    model = api.build_model("y1 = intercept + geno")
    out2 = api.runMCMC(model, ph, chain_length=20, seed=1, output_folder=str(tmp_path / "h2"), _engine=OracleEngine("block"),
                       block_size=64, output_heritability=False)
    assert "heritability" not in out2


def test_auto_sampler_follows_the_support_of_pi(tmp_path):
    """mt_bayesc_sampler_mode (MTBayesABC.jl:20-25): :auto = sampler I when Pi lists all 2^t joint states, else II."""
    d = make_dataset(n=120, p=70, ncausal=4, seed=6, center=False)
    ids = [f"id{i}" for i in range(120)]
    gdf = pd.DataFrame(d["raw"], columns=[f"snp{j}" for j in range(70)]); gdf.insert(0, "ID", ids)
    rng = np.random.default_rng(1)
    ph = pd.DataFrame({"ID": ids, "y1": d["y"], "y2": 0.5 * d["y"] + rng.standard_normal(120)})
    used = {}

    class Spy(OracleEngine):
        def init_state(self, method, t=1):
            used["method"] = method
            return super().init_state(method, t)

    for Pi, want in (({(0.0, 0.0): 0.7, (1.0, 1.0): 0.3}, "MTBayesC_II"),
                     ({(0.0, 0.0): 0.7, (1.0, 0.0): 0.1, (0.0, 1.0): 0.1, (1.0, 1.0): 0.1}, "MTBayesC")):
        geno = api.get_genotypes(gdf, np.eye(2) * 0.5, method="BayesC", Pi=Pi, multi_trait_sampler="auto")
        model = api.build_model("y1 = intercept + geno\ny2 = intercept + geno", np.eye(2))
        api.runMCMC(model, ph, chain_length=6, seed=1, output_folder=str(tmp_path / want), _engine=Spy("block"), block_size=64, outputEBV=False)
        assert used["method"] == want


def test_binary_marker_effect_samples_equal_the_text_rows(tmp_path):
    """Every saved sample is also appended as a sparse (idx, val) record to MCMC_samples_marker_effects_<geno>_<trait>.bin
    (samples.py); converting the file back gives the reference's text layout (output.jl:443-526) value for value, and the
    window GWAS reads either file."""
    from jwas_jl_amd import samples as S
    d = make_dataset(n=120, p=150, ncausal=5, seed=4, center=False)
    ids = [f"id{i}" for i in range(120)]
    gdf = pd.DataFrame(d["raw"], columns=[f"snp{j}" for j in range(150)]); gdf.insert(0, "ID", ids)
    ph = pd.DataFrame({"ID": ids, "y1": d["y"]})
    geno = api.get_genotypes(gdf, method="BayesC", Pi=0.9)
    model = api.build_model("y1 = intercept + geno")
    folder = tmp_path / "b"
    api.runMCMC(model, ph, chain_length=40, burnin=10, output_samples_frequency=3, seed=1, output_folder=str(folder),
                _engine=OracleEngine("block"), block_size=64)
    txt = folder / "MCMC_samples_marker_effects_geno_y1.txt"
    binf = folder / "MCMC_samples_marker_effects_geno_y1.bin"
    ref = pd.read_csv(txt)
    dense, mids = S.read_dense(str(binf))
    assert mids == list(ref.columns) and dense.shape == ref.shape == (10, geno.nMarkers)
    np.testing.assert_allclose(dense, ref.to_numpy(), rtol=1e-7)                  # (the text keeps 9 significant digits)
    assert (dense != 0).sum() < 0.5 * dense.size                                   # sparse records
    back = S.to_text(str(binf), str(folder / "back.txt"))
    assert open(back).read() == open(txt).read()                                   # the converter reproduces the text file
    from jwas_jl_amd.gwas import model_frequency
    pd.testing.assert_frame_equal(model_frequency(str(binf)), model_frequency(str(txt)))
    with pytest.raises(ValueError, match="not a binary marker-effect sample file"):
        S.read_dense(str(txt))


@pytest.mark.parametrize("method", ["BayesB", "BayesA"])
def test_runmcmc_multitrait_bayesb_per_marker_covariances(tmp_path, method):
    """Multi-trait BayesA/B: every marker has its own t x t effect covariance, redrawn each iteration from
    InverseWishart(df + 1, scale + b_j b_j') (variance_components.jl:181-186), and the sampler inverts each marker's matrix
    (MTBayesABC.jl:66,86-90).  Runs end to end on the CPU oracle engine; predicts held-in records of both traits."""
    from jwas_jl_amd.mcmc import _inverse_wishart_batch
    d = make_dataset(n=240, p=140, ncausal=6, seed=21, center=False)
    ids = [f"id{i}" for i in range(240)]
    gdf = pd.DataFrame(d["raw"], columns=[f"snp{j}" for j in range(140)])
    gdf.insert(0, "ID", ids)
    rng = np.random.default_rng(2)
    y1 = d["y"].astype(np.float64)
    y2 = 0.7 * y1 + 0.7 * rng.standard_normal(240)
    ph = pd.DataFrame({"ID": ids, "y1": y1, "y2": y2})
    geno = api.get_genotypes(gdf, np.eye(2) * 0.5, method=method)
    model = api.build_model("y1 = intercept + geno\ny2 = intercept + geno", np.eye(2))
    out = api.runMCMC(model, ph, chain_length=160, burnin=40, seed=4, output_folder=str(tmp_path / method),
                      _engine=OracleEngine("lookahead"), block_size=64)
    assert np.isfinite(out["residual variance"]["Estimate"]).all()
    assert not os.path.exists(tmp_path / method / "MCMC_samples_marker_effects_variances_geno.txt")   # per-marker: no common matrix
    for tr, y in (("y1", y1), ("y2", y2)):
        assert np.corrcoef(out[f"EBV_{tr}"]["EBV"].to_numpy(), y)[0, 1] > 0.5
    if method == "BayesA":                                  # every marker in the model for every trait
        me = out["marker effects geno"]
        assert (me["Model_Frequency"].to_numpy() == 1.0).all()
    # the batched inverse-Wishart draw has the right mean: E[G] = scale / (df - t - 1)
    S = np.array([[2.0, 0.5], [0.5, 1.0]])
    G = _inverse_wishart_batch(np.random.default_rng(0), 9.0, np.tile(S, (40000, 1, 1)))
    np.testing.assert_allclose(G.mean(axis=0), S / (9.0 - 2 - 1), rtol=0.03)
    with pytest.raises(ValueError, match="supported for BayesC only"):        # (the reference's own rule)
        geno = api.get_genotypes(gdf, np.eye(2) * 0.5, method="BayesB", multi_trait_sampler="II")
        model = api.build_model("y1 = intercept + geno\ny2 = intercept + geno", np.eye(2))
        api.runMCMC(model, ph, chain_length=5, seed=4, output_folder=str(tmp_path / "x"), _engine=OracleEngine("lookahead"), block_size=64)


@pytest.mark.parametrize("method", ["BayesB", "BayesA"])
def test_multitrait_bayesb_constraint_runs_mega_path_with_marker_variances(tmp_path, method):
    """constraint = true with BayesA/B: megaBayesABC! takes [vari[i,i] for vari in locus_effect_variances] (BayesABC.jl:5) and
    sample_variance(..., constraint = true) redraws only the diagonal of every marker's matrix (variance_components.jl:
    112-117,181-186): t independent single-trait BayesB chains, each marker with its own variance per trait."""
    d = make_dataset(n=240, p=140, ncausal=6, seed=21, center=False)
    ids = [f"id{i}" for i in range(240)]
    gdf = pd.DataFrame(d["raw"], columns=[f"snp{j}" for j in range(140)]); gdf.insert(0, "ID", ids)
    rng = np.random.default_rng(2)
    y1 = d["y"].astype(np.float64)
    y2 = 0.7 * y1 + 0.7 * rng.standard_normal(240)
    ph = pd.DataFrame({"ID": ids, "y1": y1, "y2": y2})
    used = {"draws": 0}

    class Spy(OracleEngine):
        def init_state(self, m, t=1):
            used["method"] = m
            return super().init_state(m, t)

        def sample_marker_covariances(self, *a, **k):
            super().sample_marker_covariances(*a, **k)
            G = self.marker_covariances()
            off = ~np.eye(2, dtype=bool)
            assert (G[:, off] == 0).all() and (G[:, ~off] > 0).all()          # diagonal draws only
            used["draws"] += 1

    geno = api.get_genotypes(gdf, np.eye(2) * 0.5, method=method, constraint=True)
    model = api.build_model("y1 = intercept + geno\ny2 = intercept + geno", np.eye(2), constraint=True)
    out = api.runMCMC(model, ph, chain_length=120, burnin=30, seed=4, output_folder=str(tmp_path / method),
                      _engine=Spy("lookahead"), block_size=64)
    assert used["method"] == "MegaBayesB" and used["draws"] == 120
    for tr, y in (("y1", y1), ("y2", y2)):
        assert np.corrcoef(out[f"EBV_{tr}"]["EBV"].to_numpy(), y)[0, 1] > 0.5
    if method == "BayesA":
        assert (out["marker effects geno"]["Model_Frequency"].to_numpy() == 1.0).all()


def test_multitrait_bayesb_restricted_support_runs_sampler_II(tmp_path):
    """mt_bayesc_sampler_mode (MTBayesABC.jl:20-25) under :auto sends a BayesB analysis whose Pi lists fewer than 2^t states
    to Gibbs sampler II (MTBayesABC.jl:129-210) with locus_effect_variances = one matrix per marker: a locus then affects
    both traits or none, and the effects follow the causal markers."""
    d = make_dataset(n=240, p=140, ncausal=6, seed=21, center=False)
    ids = [f"id{i}" for i in range(240)]
    gdf = pd.DataFrame(d["raw"], columns=[f"snp{j}" for j in range(140)]); gdf.insert(0, "ID", ids)
    rng = np.random.default_rng(2)
    y1 = d["y"].astype(np.float64)
    ph = pd.DataFrame({"ID": ids, "y1": y1, "y2": 0.7 * y1 + 0.7 * rng.standard_normal(240)})
    used = {}

    class Spy(OracleEngine):
        def init_state(self, method, t=1):
            used["method"] = method
            return super().init_state(method, t)

    geno = api.get_genotypes(gdf, np.eye(2) * 0.5, method="BayesB", Pi={(0.0, 0.0): 0.8, (1.0, 1.0): 0.2},
                             estimatePi=False, multi_trait_sampler="auto")
    model = api.build_model("y1 = intercept + geno\ny2 = intercept + geno", np.eye(2))
    out = api.runMCMC(model, ph, chain_length=120, burnin=30, seed=4, output_folder=str(tmp_path / "b2"),
                      _engine=Spy("lookahead"), block_size=64)
    assert used["method"] == "MTBayesB_II"
    me = out["marker effects geno"]
    f1 = me[me.Trait == "y1"]["Model_Frequency"].to_numpy()
    f2 = me[me.Trait == "y2"]["Model_Frequency"].to_numpy()
    assert np.array_equal(f1, f2) and 0 < f1.mean() < 1
    assert np.corrcoef(out["EBV_y1"]["EBV"].to_numpy(), y1)[0, 1] > 0.5


@pytest.mark.parametrize("fb", [50, [1, 40, 41, 150]])
def test_independent_blocks_run_the_reference_partition_too(tmp_path, fb):
    """independent_blocks = true with a block size that is not a device size, or with explicit ragged starts
    (BayesABC_block_independent!, BayesABC.jl:190-255, under JWAS.jl:298-312): the same partition as the sequential sweep."""
    n, p = 200, 230
    d = make_dataset(n=n, p=p, ncausal=3, seed=9, center=False)
    ids = [str(i) for i in range(n)]
    gdf = pd.DataFrame(d["raw"]); gdf.insert(0, "ID", ids)
    ph = pd.DataFrame({"ID": ids, "y1": d["y"]})
    geno = api.get_genotypes(gdf, method="BayesC", Pi=0.9, quality_control=False)
    model = api.build_model("y1 = intercept + geno")
    out = api.runMCMC(model, ph, chain_length=300 if np.isscalar(fb) else 6, burnin=1, fast_blocks=fb, independent_blocks=True, seed=1,
                      outputEBV=False, output_folder=str(tmp_path / "fbi"), _engine=OracleEngine("block"))
    want = list(range(1, p + 1, 50)) if np.isscalar(fb) else fb
    assert out["_timing"]["block_starts"] == want
    assert out["_timing"]["iterations"] == 6


@pytest.mark.parametrize("fb,p", [(64, 200), (50, 230), (True, 300), (7, 100)])
def test_fast_blocks_numeric_runs_the_reference_partition(tmp_path, fb, p):
    """JWAS.jl:308-312: fast_blocks = true | number cuts the markers at collect(range(1, step=block_size, stop=p)), and
    BayesABC.jl:153 repeats every block its own size (the last, shorter block fewer times): the host hands the device
    exactly that partition (uniform device blocks when the size is a device size, the ragged-partition form otherwise)."""
    n = 200
    d = make_dataset(n=n, p=p, ncausal=3, seed=9, center=False)
    ids = [str(i) for i in range(n)]
    gdf = pd.DataFrame(d["raw"]); gdf.insert(0, "ID", ids)
    ph = pd.DataFrame({"ID": ids, "y1": d["y"]})
    geno = api.get_genotypes(gdf, method="BayesC", Pi=0.9, quality_control=False)
    model = api.build_model("y1 = intercept + geno")
    s = int(np.floor(np.sqrt(n))) if fb is True else int(fb)
    out = api.runMCMC(model, ph, chain_length=6 * s, burnin=1, fast_blocks=fb, seed=1, outputEBV=False,
                      output_folder=str(tmp_path / "fb"), _engine=OracleEngine("block"))
    assert out["_timing"]["block_starts"] == list(range(1, p + 1, s))         # collect(range(1, step=s, stop=p))
    assert out["_timing"]["block_repetitions"] == 0                            # every block its own size
    assert out["_timing"]["iterations"] == 6                                   # chain_length / block_size


def test_fast_blocks_above_the_device_limit_falls_back_to_the_nearest_legal_partition(tmp_path, capsys):
    """A block size above the device's 1024-marker limit (JWAS.jl:293-316 accepts any) runs the reference's schedule at 1024
    markers per block, and an explicit start vector with an oversized block runs with that block cut into pieces -- both with
    a printed notice, never a NotImplementedError."""
    d = make_dataset(n=100, p=2500, ncausal=3, seed=9, center=False)
    ids = [str(i) for i in range(100)]
    gdf = pd.DataFrame(d["raw"]); gdf.insert(0, "ID", ids)
    ph = pd.DataFrame({"ID": ids, "y1": d["y"]})
    geno = api.get_genotypes(gdf, method="BayesC", Pi=0.9, quality_control=False)
    model = api.build_model("y1 = intercept + geno")
    out = api.runMCMC(model, ph, chain_length=2048, fast_blocks=1200, output_folder=str(tmp_path / "x"), outputEBV=False,
                      _engine=OracleEngine("block"))
    assert "exceeds the device limit of 1024" in capsys.readouterr().out
    assert out["_timing"]["iterations"] == 2                                   # chain_length / 1024
    assert out["_timing"]["block_starts"] == [1, 1025, 2049]
    geno = api.get_genotypes(gdf, method="BayesC", Pi=0.9, quality_control=False)
    model = api.build_model("y1 = intercept + geno")
    out = api.runMCMC(model, ph, chain_length=3, fast_blocks=[1, 301, 1601], output_folder=str(tmp_path / "y"), outputEBV=False,
                      _engine=OracleEngine("block"))
    assert "cut into balanced pieces" in capsys.readouterr().out
    assert out["_timing"]["block_starts"] == [1, 301, 951, 1601]               # the 1300-marker block in two halves of 650 (no 276-marker sliver)
    assert out["_timing"]["iterations"] == 3                                   # explicit starts: chain length as given


def test_runmcmc_double_precision_host_loop(tmp_path):
    """runMCMC(double_precision=true) (JWAS.jl:349-366): genotypes, residual, effects and variances stay Float64 through the
    host loop (here on the Float64 CPU oracle engine); single- and two-trait chains run and recover the signal; combinations
    the Float64 device context does not run are explicit errors."""
    from oracle_engine import OracleEngine64
    d = make_dataset(n=220, p=260, ncausal=5, seed=8, center=False)
    ids = [str(i) for i in range(220)]
    gdf = pd.DataFrame(d["raw"]); gdf.insert(0, "ID", ids)
    rng = np.random.default_rng(1)
    ph = pd.DataFrame({"ID": ids, "y1": d["y"], "y2": 0.6 * d["y"] + 0.8 * rng.standard_normal(220)})
    geno = api.get_genotypes(gdf, method="BayesC", Pi=0.9, double_precision=True)
    assert geno.genotypes.dtype == np.float64
    model = api.build_model("y1 = intercept + geno")
    out = api.runMCMC(model, ph, chain_length=80, burnin=10, seed=2, double_precision=True, output_folder=str(tmp_path / "st"),
                      _engine=OracleEngine64())
    assert out["marker effects geno"]["Estimate"].dtype == np.float64
    assert np.corrcoef(out["EBV_y1"]["EBV"], ph["y1"])[0, 1] > 0.5
    geno = api.get_genotypes(gdf, np.eye(2) * 0.5, method="BayesC", double_precision=True)      # (build_model resolves `geno` by name)
    model2 = api.build_model("y1 = intercept + geno\ny2 = intercept + geno", np.eye(2))
    out2 = api.runMCMC(model2, ph, chain_length=60, burnin=10, seed=2, double_precision=True, output_folder=str(tmp_path / "mt"),
                       _engine=OracleEngine64())
    assert np.isfinite(out2["residual variance"]["Estimate"]).all()
    # round 4: any fast_blocks partition, independent blocks and residual weights run in Float64 mode too (JWAS.jl:349-366 casts
    # everything and every mode keeps working); what stays an explicit error is sampler II / constraint / multi-trait BayesA/B
    out3 = api.runMCMC(model, ph, chain_length=100, double_precision=True, fast_blocks=50, seed=2, output_folder=str(tmp_path / "e1"),
                       _engine=OracleEngine64())
    assert out3["_timing"]["iterations"] == 2 and out3["marker effects geno"]["Estimate"].dtype == np.float64
    out4 = api.runMCMC(model, ph, chain_length=100, double_precision=True, fast_blocks=True, independent_blocks=True, seed=2,
                       output_folder=str(tmp_path / "e1b"), _engine=OracleEngine64())
    assert out4["_timing"]["block_starts"] == list(range(1, 261, 14))                       # floor(sqrt(220)) = 14
    ph_w = ph.assign(weights=1.0 + rng.uniform(0, 1, 220))
    out5 = api.runMCMC(model, ph_w, chain_length=30, double_precision=True, heterogeneous_residuals=True, seed=2,
                       output_folder=str(tmp_path / "e2"), _engine=OracleEngine64())
    assert np.isfinite(out5["residual variance"]["Estimate"]).all()
    geno = api.get_genotypes(gdf, np.eye(2) * 0.5, method="BayesC", double_precision=True, constraint=True)
    model2c = api.build_model("y1 = intercept + geno\ny2 = intercept + geno", np.eye(2))
    with pytest.raises(NotImplementedError, match="double_precision=true"):
        api.runMCMC(model2c, ph, chain_length=10, double_precision=True, output_folder=str(tmp_path / "e2c"), _engine=OracleEngine64())
    geno = api.get_genotypes(gdf, method="BayesC", Pi=0.9, double_precision=True)
    model = api.build_model("y1 = intercept + geno")
    with pytest.raises(NotImplementedError, match="runMCMC\\(double_precision=true\\)"):
        api.runMCMC(model, ph, chain_length=10, output_folder=str(tmp_path / "e3"), _engine=OracleEngine64())      # Float64 genotypes, Float32 run


def test_multitrait_host_policies():
    """Host policies of a multi-trait sampler-I chain that starts dense (mcmc.pick_block_size_mt, engine.SectionSolvePolicy):
    256-marker blocks while at least a quarter of the markers change per sweep, 512 afterwards; jwas_sweep_params.section_solve
    on while the last sweep solved at least half of its sections, off for `probe` sweeps otherwise, then tried again."""
    from jwas_jl_amd.mcmc import pick_block_size_mt
    from jwas_jl_amd.engine import SectionSolvePolicy
    assert pick_block_size_mt(99_000, 100_000) == 256 and pick_block_size_mt(10_000, 100_000) == 256      # (round 6: the dense walk down to 10 % turnover)
    assert pick_block_size_mt(9_999, 100_000) == 512 and pick_block_size_mt(100, 100_000) == 512
    # (round 6, skip and verify: 1024-marker blocks once fewer than 0.5 % of the markers change -- where the block's draws fit LDS)
    from jwas_jl_amd.mcmc import mt_1024_allowed
    assert pick_block_size_mt(100, 100_000, allow_1024=True) == 1024 and pick_block_size_mt(499, 100_000, allow_1024=True) == 1024
    assert pick_block_size_mt(500, 100_000, allow_1024=True) == 512 and pick_block_size_mt(20_000, 100_000, allow_1024=True) == 256
    assert mt_1024_allowed(3, 100_000) and mt_1024_allowed(2, 100_000) and not mt_1024_allowed(4, 100_000)
    assert not mt_1024_allowed(3, 100_000, per_marker_cov=True) and not mt_1024_allowed(3, 4096)

    class Eng:
        def __init__(self): self.solved = 0
        def last_sweep_counters(self): return [0] * 16 + [self.solved] + [0] * 7

    e = Eng()
    pol = SectionSolvePolicy(True, nsections=1560, probe=50)
    used = []
    for it in range(1, 261):
        on = pol.use(it)
        used.append(on)
        e.solved = 1560 if it <= 20 else (700 if on else 0)        # from sweep 21 on fewer than half of the sections are solved
        pol.observe(it, e)
    assert all(used[:21])                                           # on, including the sweep that found out
    assert not any(used[21:71]) and used[71]                        # off for 50 sweeps, then one probe ...
    assert not any(used[72:172]) and used[172]                      # ... which fails again: back-off, 100 sweeps until the next probe (round 6)
    off = SectionSolvePolicy(False, 1560)
    assert not off.use(1)
    off.observe(1, e)                                               # (a no-op)

    class NoCounters:                                               # the CPU oracle engine of the host tests has no counters
        pass
    pol2 = SectionSolvePolicy(True, 1560)
    pol2.observe(1, NoCounters())
    assert pol2.use(2)


def test_grouped_launch_policy():
    """Host policy of grouped launches (mcmc.grouped_blocks_for_chain / grouped_launch_size): by chain length, on the large block size
    of single-trait sparse chains only."""
    from jwas_jl_amd import mcmc as M
    assert M.grouped_blocks_for_chain(100) == 0 and M.grouped_blocks_for_chain(2999) == 0
    assert M.grouped_blocks_for_chain(3000) == 2 and M.grouped_blocks_for_chain(4999) == 2
    assert M.grouped_blocks_for_chain(5000) == M.GROUPED_BLOCKS_PER_LAUNCH == 4
    # ping-pong pairs on the 512-marker sweeps: the chains that stay in the high-turnover regime (BayesR, a fixed pi)
    assert M.pingpong_pairs_for_chain("BayesR", True, 300) == 2 and M.pingpong_pairs_for_chain("BayesC", False, 1000) == 2
    assert M.pingpong_pairs_for_chain("BayesC", False, 2000) == 4 and M.pingpong_pairs_for_chain("BayesR", True, 10 ** 6) == 4
    assert M.pingpong_pairs_for_chain("BayesC", True, 10 ** 6) == 0 and M.pingpong_pairs_for_chain("BayesR", True, 299) == 0
    assert M.pick_block_size(5500, 600_000) == 1024 and M.pick_block_size(5500, 600_000, pairs=True) == 512
    for method in ("BayesC", "BayesB", "BayesA", "BayesR"):
        assert M.grouped_launch_size(method, 1, False, 1024, 4) == 1024
        assert M.grouped_launch_size(method, 1, False, 512, 4) == 512
    assert M.grouped_launch_size("BayesC", 1, False, 1024, 0) == 0          # off
    assert M.grouped_launch_size("BayesC", 1, False, 1024, 3) == 0          # 2 or 4
    assert M.grouped_launch_size("BayesC", 3, False, 1024, 4) == 0          # multi-trait
    assert M.grouped_launch_size("BayesC", 1, True, 1024, 4) == 0           # row shards
    assert M.grouped_launch_size("BayesC", 1, False, 256, 4) == 0           # small blocks: the sampler is the critical path there
    assert M.grouped_launch_size("BayesC", 1, False, 1024, 4, dense_prior=True) == 0
    assert M.grouped_launch_size("MTBayesC", 1, False, 1024, 4) == 0
