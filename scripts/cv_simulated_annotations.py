"""The reference's 5-fold cross-validation benchmark on its packaged simulated_annotations dataset, through this package.

Protocol = benchmarks/simulated_annotations_multitrait_comparison.jl, cv mode (run_case :703-813, cv_fold_assignments
:142-154, masked_phenotype_frame :156-169): seeds 101 and 202, 5 folds, chain_length 1500, burnin 500,
output_samples_frequency 50, start h2 0.5, Pi as in the script (:8-15), estimatePi=true, quality_control=false,
center=false; held-out cor(y, EBV) and RMSE per fold.  The reference's published results for these variants
(benchmarks/reports/2026-04-11-simulated-annotations-cv-report.md) are printed next to ours.  The fold partitions
differ (Julia's MersenneTwister shuffle is not reproducible here), so agreement is statistical.

usage: python scripts/cv_simulated_annotations.py [--variants BayesC BayesR AnnotatedBayesC AnnotatedBayesR MT_I MT_II] [--json out.json]"""
import argparse
import json
import os
import sys
import tempfile
import time

import numpy as np
import pandas as pd

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import jwas_jl_amd as J  # noqa: E402
from jwas_jl_amd import api  # noqa: E402

DATA = os.path.join(ROOT, "tests", "golden", "simulated_annotations")
MT_START_PI = {(0.0, 0.0): 0.96, (1.0, 0.0): 0.015, (0.0, 1.0): 0.015, (1.0, 1.0): 0.01}
ST_BAYESC_PI = 0.98
ST_BAYESR_PI = [0.99, 0.006, 0.003, 0.001]
# held-out cor(y, EBV), trait mean over y1 and y2 (report: "Multi-Trait Family Summary", "Single-Trait Family Summary")
REFERENCE = {"BayesC": 0.6424, "BayesR": 0.6497, "MT_I": 0.6397, "MT_II": 0.6423, "AnnotatedBayesC": 0.6469, "AnnotatedBayesR": 0.6484,
             "MT_Annotated_I": 0.6338, "MT_Annotated_II": 0.6522}
REFERENCE_RMSE = {"BayesC": 4.5294, "BayesR": 4.4392, "MT_I": 5.0015, "MT_II": 4.6572, "AnnotatedBayesC": 4.6178, "AnnotatedBayesR": 4.3353,
                  "MT_Annotated_I": 4.5087, "MT_Annotated_II": 4.4700}


def folds_for(ids, nfolds, seed):
    rng = np.random.default_rng(seed)
    sh = rng.permutation(sorted(ids))
    return {i: (k % nfolds) + 1 for k, i in enumerate(sh)}


def run_variant(variant, pheno, seed, fold_of, fold, chain_length, burnin, freq, tmp, engine_factory=None):
    held = [i for i in pheno["ID"] if fold_of[i] == fold]
    mask = pheno["ID"].isin(held)
    rows = []
    if variant.startswith("MT_"):
        ymat = pheno[["y1", "y2"]].to_numpy(dtype=np.float64)
        start_g = np.cov(ymat.T) * 0.5
        start_r = np.cov(ymat.T) * 0.5
        run = pheno.copy()
        run.loc[mask, ["y1", "y2"]] = np.nan
        akw = {}
        if "Annotated" in variant:
            akw["annotations"] = pd.read_csv(os.path.join(DATA, "annotations_mt.csv")).iloc[:, 1:].to_numpy(dtype=np.float64)
        bench_geno = api.get_genotypes(os.path.join(DATA, "genotypes.csv"), start_g, separator=",", method="BayesC",
                                       estimatePi=True, quality_control=False, center=False,
                                       multi_trait_sampler="II" if variant.endswith("_II") else "I", Pi=dict(MT_START_PI), **akw)
        model = api.build_model("y1 = intercept + bench_geno\ny2 = intercept + bench_geno", start_r,
                                genotypes={"bench_geno": bench_geno})
        traits = ["y1", "y2"]
    else:
        method, trait = variant.split("_")
        annotated = method.startswith("Annotated")
        method = method.replace("Annotated", "")
        start_g = float(np.var(pheno[trait].to_numpy(dtype=np.float64), ddof=1)) * 0.5
        start_r = start_g
        run = pheno[["ID", trait]].copy()
        run.loc[mask, trait] = np.nan
        kw = dict(Pi=ST_BAYESC_PI) if method == "BayesC" else dict(Pi=list(ST_BAYESR_PI), G_is_marker_variance=False)
        if annotated:                                   # annotation_mode = :real: every column of annotations_mt.csv but the id
            kw["annotations"] = pd.read_csv(os.path.join(DATA, "annotations_mt.csv")).iloc[:, 1:].to_numpy(dtype=np.float64)
        bench_geno = api.get_genotypes(os.path.join(DATA, "genotypes.csv"), start_g, separator=",", method=method,
                                       estimatePi=True, quality_control=False, center=False, **kw)
        model = api.build_model(f"{trait} = intercept + bench_geno", start_r, genotypes={"bench_geno": bench_geno})
        traits = [trait]
    api.outputEBV(model, list(pheno["ID"]))
    t0 = time.time()
    extra = {"engine": engine_factory()} if engine_factory else {}
    out = api.runMCMC(model, run, chain_length=chain_length, burnin=burnin, output_samples_frequency=freq,
                      output_folder=os.path.join(tmp, f"{variant}_{seed}_{fold}"), seed=seed, outputEBV=True,
                      printout_model_info=False, printout_frequency=chain_length + 1, **extra)
    dt = time.time() - t0
    for tr in traits:
        ebv = out[f"EBV_{tr}"].set_index("ID")["EBV"]
        y = pheno.set_index("ID")[tr]
        e, yy = ebv.loc[held].to_numpy(dtype=np.float64), y.loc[held].to_numpy(dtype=np.float64)
        rows.append(dict(variant=variant, trait=tr, seed=seed, fold=fold, n=len(held), cor=float(np.corrcoef(e, yy)[0, 1]),
                         rmse=float(np.sqrt(np.mean((e - yy) ** 2))), seconds=dt))
    return rows


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--variants", nargs="+", default=["BayesC", "BayesR", "AnnotatedBayesC", "AnnotatedBayesR", "MT_I", "MT_II", "MT_Annotated_I", "MT_Annotated_II"])
    ap.add_argument("--seeds", type=int, nargs="+", default=[101, 202])
    ap.add_argument("--folds", type=int, default=5)
    ap.add_argument("--chain-length", type=int, default=1500)
    ap.add_argument("--burnin", type=int, default=500)
    ap.add_argument("--freq", type=int, default=50)
    ap.add_argument("--json", default=None)
    a = ap.parse_args()
    pheno = pd.read_csv(os.path.join(DATA, "phenotypes_mt.csv"), dtype={"ID": str})
    rows = []
    with tempfile.TemporaryDirectory() as tmp:
        for seed in a.seeds:
            fold_of = folds_for(list(pheno["ID"]), a.folds, seed)
            for fam in a.variants:
                cases = [fam] if fam.startswith("MT_") else [f"{fam}_y1", f"{fam}_y2"]
                for case in cases:
                    for fold in range(1, a.folds + 1):
                        rows += run_variant(case, pheno, seed, fold_of, fold, a.chain_length, a.burnin, a.freq, tmp)
    df = pd.DataFrame(rows)
    df["family"] = df["variant"].str.replace("_y1", "").str.replace("_y2", "")
    summ = []
    for fam, g in df.groupby("family"):
        per_trait = g.groupby("trait")[["cor", "rmse"]].mean()
        summ.append(dict(family=fam, cor=float(per_trait["cor"].mean()), rmse=float(per_trait["rmse"].mean()),
                         cor_y1=float(per_trait.loc["y1", "cor"]), cor_y2=float(per_trait.loc["y2", "cor"]),
                         se_cor=float(g["cor"].std(ddof=1) / np.sqrt(len(g))), seconds_per_fold=float(g["seconds"].mean()),
                         reference_cor=REFERENCE.get(fam), reference_rmse=REFERENCE_RMSE.get(fam)))
    sm = pd.DataFrame(summ)
    print(sm.to_string(index=False))
    if a.json:
        with open(a.json, "w") as fh:
            json.dump({"protocol": vars(a), "summary": summ, "per_fold": rows}, fh, indent=1)


if __name__ == "__main__":
    main()
