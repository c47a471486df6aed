"""Statistical check of the marker-shard (independent_blocks with one block per GPU) approximation: emulates G
shards on ONE GPU (G contexts) and compares the chain's hyper-parameter trajectory with the exact chain (G = 1).
--method BayesC (BASELINE config 2) or BayesR (config 3: 4-class Dirichlet pi, sigma^2 from (ssq, nnz) -- Pi.jl:11-17,
variance_components.jl:68-79,166-168; shard semantics = BayesR_block_independent!, BayesR.jl:195-273).  --json FILE writes the
trajectories and the stationary means."""
import argparse
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import jwas_jl_amd as J
from jwas_jl_amd.dist import shard_range

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=50000)
ap.add_argument("--p", type=int, default=600000)
ap.add_argument("--shards", type=int, nargs="+", default=[1, 8])
ap.add_argument("--iters", type=int, default=120)
ap.add_argument("--bs", type=int, default=512)
ap.add_argument("--seed", type=int, default=2026)
ap.add_argument("--method", choices=["BayesC", "BayesR"], default="BayesC")
ap.add_argument("--json", default=None)
ap.add_argument("--time-shards", action="store_true",
                help="record every shard's device sweep time (HIP events) and number of effect changes per iteration: what ONE RANK of the "
                     "sharded FULL chain does per iteration (its share of the full chain's turnover, not a standalone problem's)")
ap.add_argument("--pairs", action="store_true", help="with --time-shards: ping-pong pairs on the 512-marker blocks (mcmc.pingpong_pairs_for_chain)")
a = ap.parse_args()
n, p = a.n, a.p
GAMMA = np.array([0.0, 0.01, 0.1, 1.0])                  # JWAS.jl:12
report = {"method": a.method, "n": n, "p": p, "block_size": a.bs, "iters": a.iters, "seed": a.seed, "runs": {}}
for G in a.shards:
    rng = np.random.default_rng(a.seed)
    engs, rngs = [], []
    ncausal = max(1, p // 1000)
    causal = np.sort(rng.choice(p, size=ncausal, replace=False))
    eff = rng.standard_normal(ncausal)
    g = np.zeros(n)
    s2pq = 0.0
    for k in range(G):
        lo, hi = shard_range(p, k, G, align=a.bs)
        e = J.HipEngine(0)
        e.alloc_dense(n, hi - lo); e.synth(a.seed, kind=0, center=True, marker_offset=lo); e.setup_blocks(a.bs, "mfma")
        if a.pairs:
            e.setup_groups(2, "mfma")
        e.init_state(a.method, 1)
        at = np.zeros(hi - lo, dtype=np.float32)
        m = (causal >= lo) & (causal < hi)
        at[causal[m] - lo] = eff[m]
        e.set_state(alpha=at)
        g += e.mul_alpha().astype(np.float64)
        s2pq += float(e.xpx().astype(np.float64).sum()) / n
        if a.method == "BayesR": e.set_state(alpha=np.zeros(hi - lo), delta=np.ones(hi - lo, dtype=np.int32))
        else: e.set_state(alpha=np.zeros(hi - lo), beta=np.zeros(hi - lo), delta=np.ones(hi - lo))
        engs.append((e, lo, hi))
    g *= np.sqrt(0.5 / g.var())
    y = (1.0 + g + rng.standard_normal(n) * np.sqrt(0.5)).astype(np.float32)
    df_ = 4.0
    vary = float(np.var(y.astype(np.float64), ddof=1))
    vare = np.float32(0.5 * vary)
    if a.method == "BayesR":
        pi = np.array([0.95, 0.03, 0.015, 0.005])
        Gval = np.float32(0.5 * vary / (s2pq * float((GAMMA * pi).sum())))
    else:
        pi = 0.95
        Gval = np.float32(0.5 * vary / ((1.0 - pi) * s2pq))
    scale_e = float(vare) * (df_ - 2) / df_; scale_g = float(Gval) * (df_ - 2) / df_
    r = y.astype(np.float64).copy(); mu = 0.0
    hist = []
    shard_ms, shard_ev = [], []                            # [iteration][shard]
    for it in range(1, a.iters + 1):
        shard_ms.append([]); shard_ev.append([])
        r += mu
        mu = rng.standard_normal() * np.sqrt(float(vare) / n) + r.sum() / n
        r -= mu
        snap = r.astype(np.float32)
        dr = np.zeros(n, dtype=np.float32); nl = 0.0; ass = 0.0; cls = np.zeros(4); ssq = 0.0; nnz = 0.0
        for e, lo, hi in engs:
            e.set_residual(snap)
            if a.method == "BayesR":
                st = e.sweep(iteration=it, seed=a.seed, vare=vare, var_effect=Gval, pi_classes=pi, nreps=1, marker_offset=lo, group_launch=a.pairs)
                cls += st["class_counts"]; ssq += st["bayesr_ssq"]; nnz += st["bayesr_nnz"]
            else:
                st = e.sweep(iteration=it, seed=a.seed, vare=vare, var_effect=Gval, pi=pi, nreps=1, marker_offset=lo, group_launch=a.pairs)
                nl += st["sum_delta"][0]; ass += st["alpha_ss"][0, 0]
            shard_ms[-1].append(float(st["sweep_ms"])); shard_ev[-1].append(float(st["n_events"]))
            dr += e.get_residual() - snap
        r = (snap + dr).astype(np.float64)
        if a.method == "BayesR":
            nl = nnz
            pi = rng.dirichlet(cls + 1.0)
            Gval = np.float32((ssq + df_ * scale_g) / rng.chisquare(nnz + df_))
        else:
            pi = float(rng.beta(p - nl + 1.0, nl + 1.0))
            Gval = np.float32((np.float32(ass) + df_ * scale_g) / rng.chisquare(nl + df_))
        vare = np.float32((np.float32(r @ r) + df_ * scale_e) / rng.chisquare(n + df_))
        ghat = y.astype(np.float64) - mu - r
        cg = float(np.corrcoef(ghat, g)[0, 1])
        pi1 = float(pi[0]) if a.method == "BayesR" else float(pi)
        hist.append((nl, float(vare), pi1, float(Gval), cg) + (tuple(float(x) for x in cls) if a.method == "BayesR" else ()))
        if it <= 3 or it % 20 == 0:
            extra = f" classes={cls.astype(int).tolist()}" if a.method == "BayesR" else ""
            print(f"G={G} it{it}: in_model={nl:.0f} vare={vare:.4f} pi={pi1:.5f} varg={Gval:.3e} cor(ghat,g)={cg:.4f}{extra}", flush=True)
    h = np.array(hist[a.iters // 2:])
    print(f"G={G} second-half means: in_model={h[:, 0].mean():.1f} vare={h[:, 1].mean():.4f} pi={h[:, 2].mean():.5f} varg={h[:, 3].mean():.3e} cor={h[:, 4].mean():.4f}"
          + (f" classes={h[:, 5:9].mean(axis=0).round(1).tolist()}" if a.method == "BayesR" else ""), flush=True)
    names = ["in_model", "vare", "pi_null", "var_effect", "cor_ghat_g"] + (["class1", "class2", "class3", "class4"] if a.method == "BayesR" else [])
    H = np.array(hist)
    peak = int(np.argmax(H[:, 1]))
    report["runs"][str(G)] = {"second_half_means": {k: float(h[:, i].mean()) for i, k in enumerate(names)},
                              "second_half_sd": {k: float(h[:, i].std()) for i, k in enumerate(names)},
                              "vare_peak": {"iteration": peak + 1, "value": float(H[peak, 1])},
                              "trajectory": {k: [float(x) for x in H[:, i]] for i, k in enumerate(names)}}
    if a.time_shards:
        M, E = np.array(shard_ms), np.array(shard_ev)
        wins = [(lo_, min(lo_ + 30, a.iters)) for lo_ in range(30, a.iters, 30)]
        report["runs"][str(G)]["rank_share"] = {
            "what": "device sweep time (HIP events on the context's stream) of every emulated shard, per iteration of the SHARDED FULL chain",
            "pairs": bool(a.pairs),
            "windows": [{"iterations": [lo_ + 1, hi_], "ms_mean_over_shards": float(M[lo_:hi_].mean()), "ms_slowest_shard_mean": float(M[lo_:hi_].max(axis=1).mean()),
                         "events_per_shard_mean": float(E[lo_:hi_].mean()), "events_all_shards": float(E[lo_:hi_].sum(axis=1).mean())} for lo_, hi_ in wins]}
        for w in report["runs"][str(G)]["rank_share"]["windows"]:
            print(f"G={G} rank share, iterations {w['iterations']}: {w['ms_mean_over_shards']:.3f} ms per shard sweep (slowest shard {w['ms_slowest_shard_mean']:.3f}), "
                  f"{w['events_per_shard_mean']:.0f} changes per shard, {w['events_all_shards']:.0f} in all", flush=True)
    for e, _, _ in engs:
        e.close()
if a.json:
    import json
    os.makedirs(os.path.dirname(os.path.abspath(a.json)), exist_ok=True)
    with open(a.json, "w") as fh:
        json.dump(report, fh)
