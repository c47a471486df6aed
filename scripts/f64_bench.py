"""Throughput of the Float64 context (runMCMC(double_precision=true)) on one MI355X: BayesC sweeps over a synthetic
n x p matrix of doubles.  python scripts/f64_bench.py [n] [p]"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, ".")
import jwas_jl_amd as J

n = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
p = int(sys.argv[2]) if len(sys.argv) > 2 else 20000
rng = np.random.default_rng(1)
f = rng.uniform(0.1, 0.4, p)
X = np.empty((n, p), dtype=np.float64, order="F")
for j0 in range(0, p, 2000):
    j1 = min(p, j0 + 2000)
    X[:, j0:j1] = rng.binomial(2, f[j0:j1], size=(n, j1 - j0))
X -= X.mean(axis=0)
beta = np.zeros(p); q = rng.choice(p, p // 1000, replace=False); beta[q] = rng.standard_normal(q.size)
g = X @ beta
y = g / g.std() * np.sqrt(0.5) + rng.standard_normal(n) * np.sqrt(0.5)
out = {}
for prec in (64, 32):
    e = J.HipEngine(0, precision=prec)
    dt = np.float64 if prec == 64 else np.float32
    e.load_dense(np.asfortranarray(X, dtype=dt))
    t0 = time.time(); e.setup_blocks(int(os.environ.get("JWAS_F64_BLOCK", "512")) if prec == 64 else 512, "mfma"); setup = time.time() - t0
    e.init_state("BayesC", 1); e.set_residual((y - y.mean()).astype(dt))
    pi, ms = 0.95, []
    for it in range(1, 41):
        st = e.sweep(iteration=it, seed=3, vare=dt(0.5), var_effect=dt(0.5 / (0.05 * (2 * f * (1 - f)).sum())), pi=pi)
        pi = float(np.clip(1.0 - (st["sum_delta"][0] + 1.0) / (p + 2.0), 0.5, 0.9999))
        ms.append(st["sweep_ms"])
    out[f"f{prec}"] = {"sweep_ms_last10": float(np.mean(ms[-10:])), "setup_s": setup, "in_model": float(st["sum_delta"][0]),
                       "GBps_last10": (8 if prec == 64 else 4) * n * p / 1e6 / float(np.mean(ms[-10:]))}
    e.close()
print(json.dumps({"n": n, "p": p, **out}))
