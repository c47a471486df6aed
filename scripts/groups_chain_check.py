"""Long-chain check of grouped launches through the user API (development aid / evidence): the same runMCMC chain with one block per
launch and with 2 / 4 blocks per launch on a device-resident matrix -- the schedules differ by float32 rounding of the block
right-hand sides, so the CHAINS drift apart like any two float32 orderings, but posterior means, model frequencies and variance
components must agree within Monte-Carlo error.   python scripts/groups_chain_check.py [--n 20000 --p 100000 --iters 3000]"""
import argparse
import json
import os
import sys
import tempfile
import time

import numpy as np
import pandas as pd

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import jwas_jl_amd as J  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=20000)
ap.add_argument("--p", type=int, default=100000)
ap.add_argument("--iters", type=int, default=3000)
ap.add_argument("--burnin", type=int, default=500)
ap.add_argument("--json", default="")
a = ap.parse_args()

eng = J.HipEngine(0)
eng.alloc_dense(a.n, a.p)
eng.synth(2026, 0, True)
eng.setup_blocks(512, "mfma")
eng.init_state("BayesC")
rng = np.random.default_rng(1)
idx = rng.choice(a.p, 200, replace=False)
alpha = np.zeros(a.p, dtype=np.float32)
alpha[idx] = rng.standard_normal(200)
eng.set_state(alpha=alpha)
g = eng.mul_alpha()
y = (g / g.std() + rng.standard_normal(a.n)).astype(np.float32)
res = {}
for m in (0, 2, 4):
    geno = J.device_genotypes(eng, method="BayesC", Pi=0.99, estimatePi=True)
    model = J.build_model("y = intercept + geno", genotypes={"geno": geno})
    ph = pd.DataFrame({"ID": geno.obsID, "y": y})
    folder = tempfile.mkdtemp(prefix="jwas_groups_")
    t0 = time.time()
    out = J.runMCMC(model, ph, chain_length=a.iters, burnin=a.burnin, seed=7, outputEBV=False, output_samples_frequency=10,
                    output_folder=os.path.join(folder, "r"), printout_model_info=False, blocks_per_launch=m)
    me = out["marker effects geno"]
    res[m] = dict(est=np.asarray(me["Estimate"], dtype=np.float64), freq=np.asarray(me["Model_Frequency"], dtype=np.float64),
                  vare=float(out["residual variance"]["Estimate"][0]), seconds=time.time() - t0)
    print(f"blocks per launch {m or 1}: {res[m]['seconds']:.1f} s, residual variance {res[m]['vare']:.5f}, "
          f"markers with model frequency > 0.5: {(res[m]['freq'] > 0.5).sum()}", flush=True)
rep = {"n": a.n, "p": a.p, "iters": a.iters, "burnin": a.burnin}
for m in (2, 4):
    rep[f"groups{m}"] = {
        "cor_posterior_means": float(np.corrcoef(res[0]["est"], res[m]["est"])[0, 1]),
        "cor_model_frequency": float(np.corrcoef(res[0]["freq"], res[m]["freq"])[0, 1]),
        "max_abs_diff_model_frequency": float(np.abs(res[0]["freq"] - res[m]["freq"]).max()),
        "residual_variance": [res[0]["vare"], res[m]["vare"]],
        "cor_with_true_effects": [float(np.corrcoef(alpha, res[0]["est"])[0, 1]), float(np.corrcoef(alpha, res[m]["est"])[0, 1])],
        "seconds": [res[0]["seconds"], res[m]["seconds"]],
    }
print(json.dumps(rep, indent=1))
if a.json:
    json.dump(rep, open(a.json, "w"), indent=1)
eng.close()
