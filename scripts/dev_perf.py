"""Development timing probe (not the contract bench): sweep time vs block size at a given n x p."""
import argparse
import sys
import time
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import jwas_jl_amd as J

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=50000)
ap.add_argument("--p", type=int, default=40960)
ap.add_argument("--bs", type=int, nargs="+", default=[256])
ap.add_argument("--sweeps", type=int, default=6)
ap.add_argument("--pi", type=float, default=0.95)
ap.add_argument("--method", default="BayesC")
ap.add_argument("--gram", default="mfma")
ap.add_argument("--kind", type=int, default=0)
a = ap.parse_args()

e = J.HipEngine(0)
print(e.device_info(), flush=True)
t0 = time.time(); e.alloc_dense(a.n, a.p); e.synth(2026, a.kind, True); print("synth s", time.time() - t0, flush=True)
rng = np.random.default_rng(1)
for bs in a.bs:
    t0 = time.time(); e.setup_blocks(bs, a.gram); ts = time.time() - t0
    flops = 2.0 * a.p * bs * e.layout()["ld"] * (0.5 + 0.5 * 64 / bs if bs > 64 else 1.0)
    print(f"bs={bs} setup {ts:.2f}s (~{flops / ts / 1e12:.1f} TF effective gram)", flush=True)
    e.init_state(a.method)
    y = rng.standard_normal(a.n).astype(np.float32)
    e.set_residual(y)
    if a.method == "BayesR":
        e.set_state(delta=np.ones(a.p, dtype=np.int32))
    xpx = e.xpx()
    varg = np.float32(0.5 / ((1 - a.pi if a.pi < 1 else 1.0) * xpx.mean() / a.n * a.p)) if a.pi > 0 else np.float32(0.5 / (xpx.mean() / a.n * a.p))
    for it in range(1, a.sweeps + 1):
        t0 = time.time()
        if a.method == "BayesR":
            st = e.sweep(iteration=it, seed=1, vare=np.float32(1.0), var_effect=np.float32(varg * 20), pi_classes=[0.95, 0.03, 0.015, 0.005])
        else:
            st = e.sweep(iteration=it, seed=1, vare=np.float32(1.0), var_effect=varg, pi=a.pi)
        wall = (time.time() - t0) * 1e3
        gb = 4.0 * e.layout()["ld"] * a.p / 1e9
        print(f"  it{it}: sweep_ms={st['sweep_ms']:.2f} wall_ms={wall:.2f} events={st['n_events']:.0f} sum_delta={st['sum_delta']} "
              f"-> {gb / (st['sweep_ms'] * 1e-3):.0f} GB/s algorithmic", flush=True)
e.close()
