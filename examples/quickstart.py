"""Quick start: the JWAS call sequence (get_genotypes / build_model / runMCMC) on the MI355X marker path.

    python examples/quickstart.py [--n 2000 --p 20000 --method BayesC|BayesR --stream]

Mirrors the reference's README example (single-trait genomic prediction with marker effects); needs an MI355X and the
built library (jwas.jl_amd/csrc/build.sh).  --stream writes the genotypes in the reference's 2-bit packed streaming format
first (prepare_streaming_genotypes) and runs with storage="stream": same chain, 16x less HBM.
"""
import argparse
import os
import sys
import tempfile
import time

import numpy as np
import pandas as pd

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import jwas_jl_amd as J                                                     # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=2000)
ap.add_argument("--p", type=int, default=20000)
ap.add_argument("--method", default="BayesC")
ap.add_argument("--chain-length", type=int, default=1000)
ap.add_argument("--stream", action="store_true")
a = ap.parse_args()

rng = np.random.default_rng(1)
freq = rng.uniform(0.1, 0.4, a.p)
raw = (rng.random((a.n, a.p)) < freq).astype(np.float32) + (rng.random((a.n, a.p)) < freq).astype(np.float32)
qtl = rng.choice(a.p, 50, replace=False)
g = (raw - raw.mean(axis=0)) [:, qtl] @ rng.standard_normal(50)
y = 10.0 + g / g.std() * np.sqrt(0.5) + rng.standard_normal(a.n) * np.sqrt(0.5)
ids = [f"a{i}" for i in range(a.n)]
phenotypes = pd.DataFrame({"ID": ids, "y1": y.astype(np.float32)})
genotypes = pd.DataFrame(raw, columns=[f"m{j}" for j in range(a.p)])
genotypes.insert(0, "ID", ids)

with tempfile.TemporaryDirectory() as tmp:
    if a.stream:
        prefix = J.prepare_streaming_genotypes(raw, os.path.join(tmp, "geno"), obs_ids=ids, marker_ids=list(genotypes.columns[1:]))
        geno = J.get_genotypes(prefix, method=a.method, Pi=0.99 if a.method == "BayesC" else 0.0, estimatePi=True, storage="stream")
    else:
        geno = J.get_genotypes(genotypes, method=a.method, Pi=0.99 if a.method == "BayesC" else 0.0, estimatePi=True)
    model = J.build_model("y1 = intercept + geno")
    t0 = time.time()
    out = J.runMCMC(model, phenotypes, chain_length=a.chain_length, burnin=a.chain_length // 5, seed=2026,
                    output_folder=os.path.join(tmp, "results"))
    wall = time.time() - t0

me = out["marker effects geno"]
top = me.reindex(me["Model_Frequency"].sort_values(ascending=False).index).head(10)
print(top[["Marker_ID", "Estimate", "SD", "Model_Frequency"]].to_string(index=False))
print("QTL among the 50 markers with the highest model frequency:",
      len(set(me.reindex(me["Model_Frequency"].sort_values(ascending=False).index)["Marker_ID"].head(50)) & {f"m{j}" for j in qtl}))
print("residual variance:", float(out["residual variance"]["Estimate"][0]), " pi:", out["pi_geno"]["Estimate"].to_numpy())
print("cor(EBV, y):", float(np.corrcoef(out["EBV_y1"]["EBV"], phenotypes["y1"])[0, 1]))
t = out["_timing"]
print(f"{t['iterations']} iterations in {wall:.1f} s wall; device sweeps {t['device_sweep_ms_total'] / t['iterations']:.3f} ms each")
