"""Generates tests/golden/sweep_golden.npz: seeded inputs and the CPU oracle's outputs for a handful of
sweeps (BayesC at two block sizes, BayesR, 2-trait BayesC samplers I and II, 2-trait megaBayesABC).  The reference ships no golden
vectors for sampler output and cannot run here (no Julia), so these vectors pin the ORACLE's behaviour
(regression) and give the GPU box a checker-independent target; what ties the oracle to the reference
is tests/test_oracle_kat.py.    Run:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path[:0] = [os.path.join(HERE, "..", ".."), os.path.join(HERE, "..", "..", "oracle"), os.path.join(HERE, "..")]
from conftest import make_dataset  # noqa: E402
from oracle_engine import OracleEngine  # noqa: E402

CASES = {
    "bayesc_b64": dict(method="BayesC", bs=64, sweeps=12, kw=dict(vare=np.float32(0.5), var_effect=np.float32(0.004), pi=0.9)),
    "bayesc_b256": dict(method="BayesC", bs=256, sweeps=12, kw=dict(vare=np.float32(0.5), var_effect=np.float32(0.004), pi=0.9)),
    "bayesr_b64": dict(method="BayesR", bs=64, sweeps=12, kw=dict(vare=np.float32(0.5), var_effect=np.float32(0.05),
                                                                    pi_classes=np.array([0.95, 0.03, 0.015, 0.005]))),
    "mt2_b64": dict(method="MTBayesC", bs=64, sweeps=8, t=2,
                    kw=dict(vare=np.array([[0.5, 0.1], [0.1, 0.4]], dtype=np.float32),
                            var_effect=np.array([[0.004, 0.001], [0.001, 0.003]], dtype=np.float32),
                            log_prior_states=np.log(np.array([0.6, 0.1, 0.1, 0.2])))),
    "mt2_sampler2_b64": dict(method="MTBayesC_II", bs=64, sweeps=8, t=2,
                             kw=dict(vare=np.array([[0.5, 0.1], [0.1, 0.4]], dtype=np.float32),
                                     var_effect=np.array([[0.004, 0.001], [0.001, 0.003]], dtype=np.float32),
                                     log_prior_states=np.log(np.array([0.7, 0.05, 0.05, 0.2])))),
    # Rule T (jwas_sweep_params.section_solve): one full 256-marker block (four sections solved with the per-sweep inverses; a
    # prior that lets markers leave: exceptions inside the solved sections) and a ragged tail that is walked
    "mt3_rule_t_b256": dict(method="MTBayesC", bs=256, sweeps=8, t=3,
                            kw=dict(vare=np.array([[0.5, 0.1, 0.05], [0.1, 0.4, 0.08], [0.05, 0.08, 0.45]], dtype=np.float32),
                                    var_effect=np.array([[0.004, 0.001, 0.0005], [0.001, 0.003, 0.0008], [0.0005, 0.0008, 0.0035]], dtype=np.float32),
                                    log_prior_states=np.log(np.array([3e-3] * 7 + [1.0]) / (1.0 + 7 * 3e-3)), section_solve=True)),
    "mega2_b128": dict(method="MegaBayesC", bs=128, sweeps=8, t=2,
                       kw=dict(vare=np.diag([0.5, 0.4]).astype(np.float32),
                               var_effect=np.diag([0.004, 0.003]).astype(np.float32), pi=np.array([0.9, 0.8]))),
}
SEED = 424242


def inputs():
    d = make_dataset(n=210, p=330, ncausal=6, seed=12345, center=False)
    raw = d["raw"].astype(np.uint8)
    y2 = (0.5 * d["y"] + np.random.default_rng(9).standard_normal(len(d["y"])).astype(np.float32) * 0.7).astype(np.float32)
    return raw, d["y"].astype(np.float32), y2


def third_trait(y1, y2):
    return (0.7 * y1 - 0.3 * y2).astype(np.float32)


def centered(raw):
    X = raw.astype(np.float32)
    return np.asfortranarray(X - X.mean(axis=0, dtype=np.float32).astype(np.float32)[None, :])


def run_case(engine, X, Y, case):
    t = case.get("t", 1)
    engine.load_dense(X)
    engine.setup_blocks(case["bs"], "f64")
    engine.init_state(case["method"], t)
    for k in range(t):
        engine.set_residual((Y[k] - Y[k].mean()).astype(np.float32), k)
        engine.set_state(k, delta=np.ones(X.shape[1], dtype=np.int32 if case["method"] == "BayesR" else np.float32))
    for it in range(1, case["sweeps"] + 1):
        st = engine.sweep(iteration=it, seed=SEED, **case["kw"])
    out = {}
    for k in range(t):
        a, b, d = engine.get_state(k)
        out[f"alpha{k}"], out[f"delta{k}"], out[f"resid{k}"] = a, d, engine.get_residual(k)
        if case["method"] != "BayesR":
            out[f"beta{k}"] = b
    out["n_events_last"] = np.float64(st["n_events"])
    return out


def main():
    raw, y1, y2 = inputs()
    X = centered(raw)
    blob = {"raw": raw, "y1": y1, "y2": y2, "seed": np.int64(SEED)}
    for name, case in CASES.items():
        res = run_case(OracleEngine("lookahead"), X, [y1, y2, third_trait(y1, y2)], case)
        for k, v in res.items():
            blob[f"{name}/{k}"] = v
    np.savez_compressed(os.path.join(HERE, "sweep_golden.npz"), **blob)
    print("wrote", os.path.join(HERE, "sweep_golden.npz"), os.path.getsize(os.path.join(HERE, "sweep_golden.npz")), "bytes")


if __name__ == "__main__":
    main()
