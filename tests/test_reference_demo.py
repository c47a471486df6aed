"""The reference's own short tests for this path (test/runtests.jl:199-330, 352-400), run through this package on the
reference's demo_7animals example data (committed as data fixtures under tests/golden/demo_7animals/).

CPU part: genotype loading / QC / error contracts (host logic).  GPU part: the short MCMC runs -- same calls, same
structural assertions as the reference's test sets (it pins no sampler values there)."""
import os

import numpy as np
import pandas as pd
import pytest

import jwas_jl_amd as J
from jwas_jl_amd import api

DEMO = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "demo_7animals")
GENO = os.path.join(DEMO, "genotypes.txt")
PHENO = os.path.join(DEMO, "phenotypes.txt")


def _phenotypes():
    return pd.read_csv(PHENO, sep=",", na_values=["NA"], dtype={"ID": str})


# ---- "Genotype Loading" (runtests.jl:198-230) -------------------------------------------------------------
def test_load_from_file_with_header():
    geno = api.get_genotypes(GENO, 1.0, separator=",", header=True, method="BayesC")
    assert geno.nMarkers > 0 and geno.nObs > 0
    assert geno.centered is True
    assert geno.method == "BayesC"
    assert len(geno.obsID) == geno.nObs == 7
    assert len(geno.markerID) == geno.nMarkers
    assert list(geno.obsID) == ["a1", "a3", "a4", "a5", "a6", "a7", "a8"]
    # centred columns; allele frequency = column mean / 2 (readgenotypes.jl:384-385)
    raw = pd.read_csv(GENO).iloc[:, 1:].to_numpy(dtype=np.float64)
    keep = [m in geno.markerID for m in ["m1", "m2", "m3", "m4", "m5"]]
    np.testing.assert_allclose(np.asarray(geno.genotypes, dtype=np.float64), (raw - raw.mean(0))[:, keep], atol=1e-6)
    np.testing.assert_allclose(np.asarray(geno.alleleFreq).ravel(), (raw.mean(0) / 2)[keep], atol=1e-6)


@pytest.mark.parametrize("method", ["BayesA", "BayesB", "BayesC", "BayesR", "RR-BLUP", "BayesL"])
def test_load_with_device_methods(method):
    geno = api.get_genotypes(GENO, 1.0, separator=",", method=method)
    assert geno.method == method and geno.nMarkers > 0


@pytest.mark.parametrize("method", ["GBLUP"])
def test_methods_off_the_device_path_are_refused_loudly(method):
    """The reference loads these too (runtests.jl:213-220); they are outside the hot path built here and must not
    silently fall back to anything."""
    with pytest.raises(NotImplementedError, match="not on the device path"):
        api.get_genotypes(GENO, 1.0, separator=",", method=method)


def test_quality_control_never_adds_markers():
    with_qc = api.get_genotypes(GENO, 1.0, separator=",", quality_control=True, MAF=0.01)
    no_qc = api.get_genotypes(GENO, 1.0, separator=",", quality_control=False)
    assert with_qc.nMarkers <= no_qc.nMarkers == 5


# ---- "Edge Cases" (runtests.jl:352-366) and "Data Types" (:385-400) ----------------------------------------
def test_invalid_residual_variance():
    with pytest.raises(ValueError):
        api.build_model("y = intercept", -1.0)


def test_invalid_covariance_matrix():
    with pytest.raises(ValueError):
        api.build_model("y1 = intercept\ny2 = intercept", np.array([[1.0, 2.0], [2.0, 1.0]]))


def test_single_precision_default_and_double_on_request():
    """test/runtests.jl: genotypes are Float32 unless double_precision=true asks for Float64 (readgenotypes.jl:298)."""
    geno = api.get_genotypes(GENO, 1.0, separator=",", double_precision=False)
    assert np.asarray(geno.genotypes).dtype == np.float32
    geno64 = api.get_genotypes(GENO, 1.0, separator=",", double_precision=True)
    assert np.asarray(geno64.genotypes).dtype == np.float64
    np.testing.assert_allclose(geno64.genotypes, geno.genotypes, atol=1e-6)


# ---- "Input Validation" (test/unit/test_input_validation.jl:9-40) ---------------------------------------------
def test_invalid_bayesian_method(tmp_path):
    geno = api.get_genotypes(GENO, 1.0, separator=",", method="BayesC")
    geno.method = "InvalidMethod"
    model = api.build_model("y1 = intercept + geno", 1.0)
    with pytest.raises(ValueError, match="is not available in JWAS"):
        api.runMCMC(model, _phenotypes(), chain_length=10, output_folder=str(tmp_path / "x"), seed=123)


def test_output_samples_frequency_validation(tmp_path):
    geno = api.get_genotypes(GENO, 1.0, separator=",", method="BayesC")
    model = api.build_model("y1 = intercept + geno", 1.0)
    with pytest.raises(ValueError, match="output_samples_frequency should be an integer > 0"):
        api.runMCMC(model, _phenotypes(), chain_length=10, output_samples_frequency=0, output_folder=str(tmp_path / "x"), seed=123)


# ---- "MCMC Functionality" (runtests.jl:258-320), "Model frequency" (:333-349) -------------------------------
@pytest.mark.gpu
def test_single_trait_bayesc_short_run(tmp_path):
    geno = api.get_genotypes(GENO, 1.0, separator=",", method="BayesC")
    model = api.build_model("y1 = intercept + geno", 1.0)
    out = api.runMCMC(model, _phenotypes(), chain_length=50, burnin=10, output_samples_frequency=10,
                      output_folder=str(tmp_path / "results"), seed=123)
    assert "location parameters" in out and "residual variance" in out and "marker effects geno" in out
    assert len(out["location parameters"]) > 0
    me = out["marker effects geno"]
    assert list(me.columns) == ["Trait", "Marker_ID", "Estimate", "SD", "Model_Frequency"]     # output.jl:108-147
    assert len(me) == geno.nMarkers
    assert ((me["Model_Frequency"] >= 0) & (me["Model_Frequency"] <= 1)).all()
    assert np.isfinite(me["Estimate"]).all()
    assert os.path.isfile(tmp_path / "results" / "location_parameters.txt")
    assert os.path.isfile(tmp_path / "results" / "residual_variance.txt")
    assert os.path.isfile(tmp_path / "results" / "MCMC_samples_marker_effects_geno_y1.txt")
    samples = pd.read_csv(tmp_path / "results" / "MCMC_samples_marker_effects_geno_y1.txt")
    assert samples.shape == (4, geno.nMarkers)            # (50 - 10) / 10 saved samples x markers
    freq = (samples.to_numpy() != 0).mean(0)              # GWAS(): model frequency from the samples file
    assert ((freq >= 0) & (freq <= 1)).all()


@pytest.mark.gpu
def test_output_folder_creation_rrblup(tmp_path):
    """runtests.jl:283-297"""
    geno = api.get_genotypes(GENO, 1.0, separator=",", method="RR-BLUP")
    model = api.build_model("y1 = intercept + geno", 1.0)
    folder = tmp_path / "test_results_temp"
    out = api.runMCMC(model, _phenotypes(), chain_length=50, output_folder=str(folder), seed=123)
    assert os.path.isdir(folder)
    assert os.path.isfile(folder / "location_parameters.txt") and os.path.isfile(folder / "residual_variance.txt")
    assert (out["marker effects geno"]["Model_Frequency"] == 1.0).all()        # every marker is in the model


@pytest.mark.gpu
def test_reproducibility_with_seed(tmp_path):
    outs = []
    for tag in ("temp1", "temp2"):
        geno = api.get_genotypes(GENO, 1.0, separator=",", method="RR-BLUP")
        model = api.build_model("y1 = intercept + geno", 1.0)
        outs.append(api.runMCMC(model, _phenotypes(), chain_length=50, output_folder=str(tmp_path / tag), seed=999))
    assert abs(outs[0]["residual variance"]["Estimate"][0] - outs[1]["residual variance"]["Estimate"][0]) <= 1e-10
    np.testing.assert_array_equal(outs[0]["marker effects geno"]["Estimate"].to_numpy(),
                                  outs[1]["marker effects geno"]["Estimate"].to_numpy())


@pytest.mark.gpu
@pytest.mark.parametrize("method", ["BayesA", "BayesB", "BayesR", "BayesL", "RR-BLUP"])      # test_bayesb_methods.jl:8-54
def test_other_single_trait_methods_short_run(tmp_path, method):
    geno = api.get_genotypes(GENO, 1.0, separator=",", method=method)
    model = api.build_model("y1 = intercept + geno", 1.0)
    out = api.runMCMC(model, _phenotypes(), chain_length=50, burnin=10, output_folder=str(tmp_path / method), seed=123)
    assert len(out["marker effects geno"]) == geno.nMarkers
    assert np.isfinite(out["residual variance"]["Estimate"]).all()


@pytest.mark.gpu
@pytest.mark.parametrize("sampler", ["I", "II"])
def test_multi_trait_bayesc_short_run(tmp_path, sampler):
    """test_multitrait_mcmc.jl:111-128, 239-265: multi-trait BayesC on the demo data as it is -- a5 has y1 but no y2
    (its missing residual is imputed every iteration, residual.jl:52-73); a2 is not genotyped, a6-a8 have no records."""
    G = np.array([[1.0, 0.5], [0.5, 1.0]])
    R = np.array([[1.0, 0.5], [0.5, 1.0]])
    geno = api.get_genotypes(GENO, G, separator=",", method="BayesC", multi_trait_sampler=sampler)
    assert geno.multi_trait_sampler == sampler
    model = api.build_model("y1 = intercept + geno\ny2 = intercept + geno", R)
    out = api.runMCMC(model, _phenotypes(), chain_length=100, burnin=20, output_samples_frequency=10,
                      output_folder=str(tmp_path / "mt"), seed=123)
    assert "location parameters" in out and "residual variance" in out and "marker effects geno" in out
    assert len(out["residual variance"]) == 4                     # 2x2 covariance flattened
    me = out["marker effects geno"]
    assert len(me) == 2 * geno.nMarkers and set(me["Trait"]) == {"y1", "y2"}
    used = open(tmp_path / "mt" / "IDs_for_individuals_with_phenotypes.txt").read().split()
    assert used == ["a1", "a3", "a4", "a5"]
    assert list(out["EBV_y2"]["ID"]) == list(geno.obsID)           # EBVs for all 7 genotyped animals


def test_multi_trait_sampler_default_and_auto():
    """test_multitrait_mcmc.jl:130-138"""
    G = np.array([[1.0, 0.5], [0.5, 1.0]])
    assert api.get_genotypes(GENO, G, separator=",", method="BayesC").multi_trait_sampler == "I"
    assert api.get_genotypes(GENO, G, separator=",", method="BayesC", multi_trait_sampler="auto").multi_trait_sampler == "auto"


@pytest.mark.gpu
def test_ebv_output_with_genotypes(tmp_path):
    """test_output_ebv.jl:9-31"""
    geno = api.get_genotypes(GENO, 1.0, separator=",", method="BayesC")
    model = api.build_model("y1 = intercept + geno", 1.0)
    api.outputEBV(model, geno.obsID)
    out = api.runMCMC(model, _phenotypes(), chain_length=100, burnin=20, output_samples_frequency=10, outputEBV=True,
                      output_folder=str(tmp_path / "ebv"), seed=123)
    ebv = out["EBV_y1"]
    assert list(ebv.columns) == ["ID", "EBV", "PEV"] and len(ebv) == 7
    assert (ebv["PEV"] >= 0).all()


@pytest.mark.gpu
@pytest.mark.parametrize("sampler,independent", [("I", False), ("II", False), ("I", True), ("II", True)])
def test_multi_trait_fast_blocks_runs(tmp_path, sampler, independent):
    """test_multitrait_mcmc.jl:267-322, 373-443: multi-trait BayesC with fast_blocks=true (block size floor(sqrt(nObs)) = 2
    in the reference; 5 markers are fewer than one device block, so the device runs them as a single block with the
    reference's repetition schedule: 2 repetitions, chain_length / 2 outer iterations) and with independent blocks, samplers I and II."""
    G = np.array([[1.0, 0.5], [0.5, 1.0]])
    geno = api.get_genotypes(GENO, G, separator=",", method="BayesC", quality_control=False, multi_trait_sampler=sampler)
    model = api.build_model("y1 = intercept + geno\ny2 = intercept + geno", G)
    out = api.runMCMC(model, _phenotypes(), chain_length=100, burnin=5, output_samples_frequency=5,
                      output_folder=str(tmp_path / "fb"), seed=123, fast_blocks=True, independent_blocks=independent)
    assert "marker effects geno" in out and "location parameters" in out
    assert np.isfinite(out["marker effects geno"]["Estimate"]).all()
    assert out["_timing"]["iterations"] == 50                     # chain_length / floor(sqrt(4 records)) outer iterations


def test_fast_blocks_needs_two_block_starts(tmp_path):
    """JWAS.jl:308-311"""
    geno = api.get_genotypes(GENO, 1.0, separator=",", method="BayesC", quality_control=False)
    model = api.build_model("y1 = intercept + geno", 1.0)
    from oracle_engine import OracleEngine
    with pytest.raises(ValueError, match="at least two block starts"):
        api.runMCMC(model, _phenotypes(), chain_length=20, output_folder=str(tmp_path / "x"), seed=1, fast_blocks=7,
                    _engine=OracleEngine("block"))


# ---- test/unit/test_bayesr.jl:62-140, 282-344 ---------------------------------------------------------------
BAYESR_PI = [0.95, 0.03, 0.015, 0.005]


def test_bayesr_genotype_loads():
    geno = api.get_genotypes(GENO, 1.0, separator=",", method="BayesR", Pi=list(BAYESR_PI), estimatePi=True)
    model = api.build_model("y1 = intercept + geno", 1.0)
    assert geno.method == "BayesR" and model.nModels == 1


def test_bayesr_rejects_bad_pi_length(tmp_path):
    from oracle_engine import OracleEngine
    geno = api.get_genotypes(GENO, 1.0, separator=",", method="BayesR", Pi=[0.95, 0.05, 0.0], estimatePi=True)
    model = api.build_model("y1 = intercept + geno", 1.0)
    with pytest.raises(ValueError, match="length 4"):
        api.runMCMC(model, _phenotypes(), chain_length=10, output_folder=str(tmp_path / "x"), seed=1, _engine=OracleEngine("block"))


def test_bayesr_does_not_mutate_caller_pi(tmp_path):
    from oracle_engine import OracleEngine
    start_pi = np.array(BAYESR_PI)
    original = start_pi.copy()
    geno = api.get_genotypes(GENO, 1.0, separator=",", method="BayesR", Pi=start_pi, estimatePi=True)
    model = api.build_model("y1 = intercept + geno", 1.0)
    api.runMCMC(model, _phenotypes(), chain_length=10, burnin=0, output_samples_frequency=5, output_folder=str(tmp_path / "x"),
                seed=123, printout_model_info=False, outputEBV=False, _engine=OracleEngine("block"))
    assert np.array_equal(start_pi, original)


@pytest.mark.gpu
def test_bayesr_runs_and_estimatepi_output(tmp_path):
    geno = api.get_genotypes(GENO, 1.0, separator=",", method="BayesR", Pi=list(BAYESR_PI), estimatePi=True, estimate_variance=True)
    model = api.build_model("y1 = intercept + geno", 1.0)
    out = api.runMCMC(model, _phenotypes(), chain_length=30, burnin=5, output_samples_frequency=5, output_folder=str(tmp_path / "a"),
                      seed=321, printout_model_info=False, outputEBV=False, fast_blocks=False)
    assert len(out["pi_geno"]) == 4 and abs(out["pi_geno"]["Estimate"].sum() - 1.0) < 1e-6
    assert out["marker effects geno"]["Model_Frequency"].between(0, 1).all()
    # fast_blocks=true gets past validation and runs; fast_blocks=1 is the plain chain (block size 1, 1 repetition)
    for fb in (True, 1):
        geno = api.get_genotypes(GENO, 1.0, separator=",", method="BayesR", Pi=list(BAYESR_PI), estimatePi=False, estimate_variance=False)
        model = api.build_model("y1 = intercept + geno", 1.0)
        o = api.runMCMC(model, _phenotypes(), chain_length=10, burnin=0, output_samples_frequency=5, output_folder=str(tmp_path / f"fb{fb}"),
                        seed=123, printout_model_info=False, outputEBV=False, fast_blocks=fb)
        assert o["_timing"]["iterations"] == (5 if fb is True else 10)       # floor(sqrt(4 records)) = 2 -> 10 / 2


@pytest.mark.gpu
def test_ebv_output_with_heritability(tmp_path):
    """test_output_ebv.jl:33-55 (there with RR-BLUP)"""
    geno = api.get_genotypes(GENO, 1.0, separator=",", method="RR-BLUP")
    model = api.build_model("y1 = intercept + geno", 1.0)
    out = api.runMCMC(model, _phenotypes(), chain_length=100, burnin=20, output_samples_frequency=10, outputEBV=True,
                      output_heritability=True, output_folder=str(tmp_path / "h2"), seed=123)
    assert "EBV_y1" in out and "heritability" in out and "genetic_variance" in out
    h2 = out["heritability"]
    assert "Estimate" in h2.columns and h2["Estimate"].between(0, 1).all()
    assert os.path.isfile(tmp_path / "h2" / "MCMC_samples_heritability.txt")


def test_build_model_genotype_contracts():
    """build_MME.jl:104-116 (test_annotated_bayesc.jl:607-630; test_multitrait_mcmc.jl)"""
    geno = api.get_genotypes(GENO, 1.0, separator=",", method="BayesC", multi_trait_sampler="II")
    with pytest.raises(ValueError, match="require multi-trait BayesC"):
        api.build_model("y1 = intercept + geno", 1.0)
    geno = api.get_genotypes(GENO, np.eye(2), separator=",", method="BayesC")
    with pytest.raises(ValueError, match="not a 1 by 1 matrix"):
        api.build_model("y1 = intercept + geno", 1.0)
    with pytest.raises(ValueError, match="multi_trait_sampler must be one of"):
        api.get_genotypes(GENO, 1.0, separator=",", method="BayesC", multi_trait_sampler="bogus")


# ---- test/unit/test_advanced_coverage.jl:77-206 -------------------------------------------------------------
@pytest.mark.gpu
def test_advanced_coverage_sets_on_the_device_path(tmp_path):
    G = np.array([[1.0, 0.5], [0.5, 1.0]])
    I2 = np.eye(2)
    ph = _phenotypes()
    # mega-trait (G constraint), records with missing traits included
    geno = api.get_genotypes(GENO, I2, separator=",", method="BayesC", constraint=True)
    model = api.build_model("y1 = intercept + geno\ny2 = intercept + geno", I2)
    out = api.runMCMC(model, ph, chain_length=100, burnin=20, output_samples_frequency=10, output_folder=str(tmp_path / "c"), seed=123)
    assert "location parameters" in out and "residual variance" in out
    # multi-trait RR-BLUP with missing phenotypes
    geno = api.get_genotypes(GENO, G, separator=",", method="RR-BLUP")
    model = api.build_model("y1 = intercept + geno\ny2 = intercept + geno", G)
    out = api.runMCMC(model, ph, chain_length=100, burnin=20, output_samples_frequency=10, missing_phenotypes=True,
                      output_folder=str(tmp_path / "m"), seed=123)
    assert "location parameters" in out and "residual variance" in out
    # Pi estimation in multi-trait BayesC
    geno = api.get_genotypes(GENO, G, separator=",", method="BayesC", estimatePi=True)
    model = api.build_model("y1 = intercept + geno\ny2 = intercept + geno", G)
    out = api.runMCMC(model, ph, chain_length=100, burnin=20, output_samples_frequency=10, output_folder=str(tmp_path / "p"), seed=123)
    assert "pi_geno" in out and "marker effects geno" in out
    # single-trait BayesC with estimate_scale
    geno = api.get_genotypes(GENO, 1.0, separator=",", method="BayesC", estimate_scale=True)
    model = api.build_model("y1 = intercept + geno", 1.0)
    out = api.runMCMC(model, ph, chain_length=100, burnin=20, output_samples_frequency=10, output_folder=str(tmp_path / "s"), seed=123)
    assert "location parameters" in out and "ScaleEffectVargeno" in out and float(out["ScaleEffectVargeno"]["Estimate"][0]) > 0
    # multi-trait EBV output with heritability
    geno = api.get_genotypes(GENO, G, separator=",", method="BayesC")
    model = api.build_model("y1 = intercept + geno\ny2 = intercept + geno", G)
    api.outputEBV(model, geno.obsID)
    out = api.runMCMC(model, ph, chain_length=100, burnin=20, output_samples_frequency=10, outputEBV=True, output_heritability=True,
                      output_folder=str(tmp_path / "e"), seed=123)
    assert {"EBV_y1", "EBV_y2", "genetic_variance", "heritability"} <= set(out)
    assert len(out["genetic_variance"]) == 4 and len(out["heritability"]) == 2
    # single-trait with weights (heterogeneous residuals): the demo file has a `weights` column
    geno = api.get_genotypes(GENO, 1.0, separator=",", method="BayesC")
    model = api.build_model("y1 = intercept + geno", 1.0)
    out = api.runMCMC(model, ph, chain_length=100, burnin=20, output_samples_frequency=10, heterogeneous_residuals=True,
                      output_folder=str(tmp_path / "w"), seed=123)
    assert "marker effects geno" in out
    # multi-trait BayesB runs on the device (one effect covariance per marker); multi-trait BayesL stays on the reference, loudly
    geno = api.get_genotypes(GENO, G, separator=",", method="BayesB")
    model = api.build_model("y1 = intercept + geno\ny2 = intercept + geno", G)
    out = api.runMCMC(model, ph, chain_length=40, burnin=10, output_folder=str(tmp_path / "b"), seed=123)
    assert np.isfinite(out["marker effects geno"]["Estimate"]).all() and np.isfinite(out["residual variance"]["Estimate"]).all()
    geno = api.get_genotypes(GENO, G, separator=",", method="BayesL")
    model = api.build_model("y1 = intercept + geno\ny2 = intercept + geno", G)
    with pytest.raises(NotImplementedError, match="multi-trait"):
        api.runMCMC(model, ph, chain_length=10, output_folder=str(tmp_path / "l"), seed=123)


# ---- test/unit/test_misc_coverage.jl:69-91, 116-208 ---------------------------------------------------------
def test_output_mcmc_samples_and_covariates(tmp_path):
    from oracle_engine import OracleEngine
    geno = api.get_genotypes(GENO, 1.0, separator=",", method="BayesC")
    model = api.build_model("y1 = intercept + x1 + geno", 1.0)
    api.set_covariate(model, "x1")
    assert "x1" in model.covVec
    api.outputMCMCsamples(model, "intercept", "x1")
    folder = tmp_path / "test_multi_samples"
    api.runMCMC(model, _phenotypes(), chain_length=50, output_samples_frequency=10, output_folder=str(folder), seed=123,
                _engine=OracleEngine("block"))
    assert os.path.isfile(folder / "MCMC_samples_y1.intercept.txt") and os.path.isfile(folder / "MCMC_samples_y1.x1.txt")
    assert pd.read_csv(folder / "MCMC_samples_y1.x1.txt").shape == (5, 1)


@pytest.mark.parametrize("bad,msg", [([2, 4], "begin with 1"), ([1, 3, 3], "sorted and unique"), ([3, 1], "begin with 1"),
                                     ([1, 10], "within 1:nMarkers")])
def test_explicit_block_start_validation(tmp_path, bad, msg):
    from oracle_engine import OracleEngine
    geno = api.get_genotypes(GENO, 1.0, separator=",", method="BayesC")
    model = api.build_model("y1 = intercept + geno", 1.0)
    with pytest.raises(ValueError, match=msg):
        api.runMCMC(model, _phenotypes(), chain_length=6, output_folder=str(tmp_path / "x"), fast_blocks=bad, _engine=OracleEngine("block"))
    geno = api.get_genotypes(GENO, 1.0, separator=",", method="BayesC")
    model = api.build_model("y1 = intercept + geno", 1.0)
    # a NON-uniform start vector runs as given: blocks of 1, 2 and 2 markers, each with its own size as repetition count
    out = api.runMCMC(model, _phenotypes(), chain_length=6, output_folder=str(tmp_path / "y"), fast_blocks=[1, 2, 4], seed=3,
                      _engine=OracleEngine("block"))
    assert out["_timing"]["iterations"] == 6 and np.isfinite(out["marker effects geno"]["Estimate"]).all()


@pytest.mark.parametrize("method", ["BayesC", "BayesR"])
def test_independent_blocks_with_explicit_uniform_starts(tmp_path, method):
    """test_misc_coverage.jl:161-196: fast_blocks=[1,3,5] (blocks of 2 markers) with independent_blocks=true; the chain
    length is NOT rescaled for an explicit start vector (JWAS.jl:298-304)."""
    from oracle_engine import OracleEngine
    geno = api.get_genotypes(GENO, 1.0, separator=",", method=method)
    model = api.build_model("y1 = intercept + geno", 1.0)
    out = api.runMCMC(model, _phenotypes(), chain_length=6, output_folder=str(tmp_path / "ib"), seed=1, fast_blocks=[1, 3, 5],
                      independent_blocks=True, _engine=OracleEngine("block"))
    assert "marker effects geno" in out and "Model_Frequency" in out["marker effects geno"].columns
    assert out["_timing"]["iterations"] == 6
