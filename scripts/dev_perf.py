"""Development timing probe (not the contract bench): device sweep time of any sampler at a given n x p."""
import argparse
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import jwas_jl_amd as J

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=50000)
ap.add_argument("--p", type=int, default=40960)
ap.add_argument("--bs", type=int, nargs="+", default=[512])
ap.add_argument("--sweeps", type=int, default=12)
ap.add_argument("--pi", type=float, default=0.95)
ap.add_argument("--method", default="BayesC")      # BayesC | BayesR | MT
ap.add_argument("--traits", type=int, default=3)
ap.add_argument("--gram", default="mfma")
ap.add_argument("--kind", type=int, default=0)
ap.add_argument("--independent", action="store_true")
ap.add_argument("--packed", action="store_true")
ap.add_argument("--nreps", type=int, default=1)
ap.add_argument("--mtprior", default="corners")   # corners: only 0..0 and 1..1 | spread: (1-pi) split over all non-null states
ap.add_argument("--sampler", default="I")
a = ap.parse_args()

e = J.HipEngine(0)
t0 = time.time(); (e.alloc_packed if a.packed else e.alloc_dense)(a.n, a.p); e.synth(2026, a.kind, True); print("synth s", round(time.time() - t0, 2), flush=True)
rng = np.random.default_rng(1)
for bs in a.bs:
    t0 = time.time(); e.setup_blocks(bs, a.gram); print(f"bs={bs} setup {time.time() - t0:.2f}s", flush=True)
    t = a.traits if a.method == "MT" else 1
    e.init_state(("MTBayesC_II" if a.sampler == "II" else "MTBayesC") if a.method == "MT" else a.method, t)
    # y with a few causal markers
    at = np.zeros(a.p, dtype=np.float32); idx = rng.choice(a.p, max(1, a.p // 1000), replace=False); at[idx] = rng.standard_normal(len(idx))
    e.set_state(0, alpha=at); g = e.mul_alpha(0); g = g / g.std() * np.sqrt(0.5)
    for k in range(t):
        y = (g + rng.standard_normal(a.n) * np.sqrt(0.5)).astype(np.float32)
        e.set_state(k, alpha=np.zeros(a.p), beta=np.zeros(a.p), delta=np.ones(a.p, dtype=np.int32 if a.method == "BayesR" else np.float32))
        e.set_residual(y - y.mean(), k)
    s2pq = float(e.xpx().astype(np.float64).sum()) / a.n
    gb = 4.0 * a.n * a.p / 1e9
    pi = a.pi
    varg = np.float32(0.5 / ((1 - pi) * s2pq))
    pi4 = np.array([0.95, 0.03, 0.015, 0.005]); sig = np.float32(0.5 / (s2pq * (0.03 * 0.01 + 0.015 * 0.1 + 0.005)))
    lp = np.full(1 << t, -np.inf); lp[(1 << t) - 1] = np.log(1 - pi); lp[0] = np.log(pi)
    if a.mtprior == "spread":
        lp[1:] = np.log((1 - pi) / ((1 << t) - 1))
    xkw = dict(independent_blocks=a.independent, nreps=a.nreps)
    for it in range(1, a.sweeps + 1):
        if a.method == "BayesR":
            st = e.sweep(iteration=it, seed=1, vare=np.float32(0.5), var_effect=sig, pi_classes=pi4, **xkw)
            nin = st["class_counts"][1:].sum(); pi4 = (st["class_counts"] + 1) / (a.p + 4)
        elif a.method == "MT":
            st = e.sweep(iteration=it, seed=1, vare=(np.eye(t) * 0.5).astype(np.float32), var_effect=(np.eye(t) * varg).astype(np.float32), log_prior_states=lp, **xkw)
            nin = st["sum_delta"][0]
            pr = (st["state_counts"] + 1) / (a.p + (1 << t)); lp = np.log(pr)
        else:
            st = e.sweep(iteration=it, seed=1, vare=np.float32(0.5), var_effect=varg, pi=pi, **xkw)
            nin = st["sum_delta"][0]; pi = float(1 - (nin + 1) / (a.p + 2))
        if it <= 2 or it % 4 == 0:
            print(f"  it{it}: sweep_ms={st['sweep_ms']:.2f} events={st['n_events']:.0f} in_model={nin:.0f} -> {gb / (st['sweep_ms'] * 1e-3):.0f} GB/s algorithmic", flush=True)
e.close()
