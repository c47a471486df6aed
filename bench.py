#!/usr/bin/env python
"""bench.py -- MCMC iterations/sec of the marker-effect Gibbs sampler on MI355X.

Metric (BASELINE.json): MCMC iters/sec (full marker sweep), 50k x 600k single-trait BayesC, fp32 dense
genotypes, pi0 = 0.95 estimated, at 1/2/4/8 GPUs.  One "step" = one MCMC iteration = one full sweep
over all p markers (device) + the host-side updates of MCMC_BayesianAlphabet.jl:196-220,294-370
(intercept Gibbs step, pi ~ Beta, marker-effect variance, residual variance).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

N > 1: markers are sharded over the ranks (jwas.jl_amd/dist.py): each rank sweeps p/N markers from the
same residual snapshot and one all-reduce (RCCL) of the residual delta reconciles per sweep.  Total
work is fixed (50k x 600k), so "scaling" is "strong".

Prints ONE JSON line (rank 0) with the contract fields plus
  "roofline":     HBM roofline of the dominant kernel (k_block_step: sampler of block k-1 || update + partial
                  RHS of block k): algorithmic bytes per launch / average launch duration, from the HIP events
                  recorded on the sweep's stream inside the timed region;
  "cpu_baseline": the CPU oracle's non-block BayesC sweep (the reference's per-marker sdot/saxpy
                  order) timed on this box's host cores on a marker subsample (N = 1, rank 0 only).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

N_IND, P_TOTAL = 50_000, 600_000
HBM_PEAK_GBS = 8000.0                       # MI355X_MICROARCH.md: HBM3E 8 TB/s peak
# HBM bytes per k_block_step launch from the PMC passes committed under profiles/ (rocprofv3 --pmc FETCH_SIZE and
# --pmc WRITE_SIZE in separate runs of this command, averaged over the launches of the timed region's steady state;
# FETCH_SIZE x2 per the guide's gfx950 correction, checked on k_xpx which reads X exactly once).
# Valid for the default config only (n=50000, p=600000, adaptive blocks -> 1024 in the timed region): 213.7 MB per launch
# against 204.4 MB algorithmic (1.05x).
TRAFFIC_BYTES_PER_LAUNCH = (101283.94 * 2 + 655.96) * 1024          # profiles/r01_pmc_{fetch,write}_summary.csv, last rows


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=30)
    ap.add_argument("--burnin", type=int, default=30,
                    help="chain iterations run as part of the SETUP (untimed, before the warm-up steps): the metric is defined "
                         "on the steady state of the chain (SURVEY.md section 8d), which a start from alpha = 0 reaches after "
                         "~25 sweeps; with --warmup >= 30 (the default) pass --burnin 0 for the same state")
    ap.add_argument("--n", type=int, default=N_IND)
    ap.add_argument("--p", type=int, default=P_TOTAL)
    ap.add_argument("--block-size", type=int, default=int(os.environ.get("JWAS_BLOCK_SIZE", "0")),
                    help="0 (default) = the host loop's policy (jwas.jl_amd/mcmc.py): blocks of 512 markers while many effects "
                         "change per sweep, 1024 once fewer than 1.25 %% do; or a fixed size in {64,...,1024}")
    ap.add_argument("--seed", type=int, default=2026)
    ap.add_argument("--workload", choices=["config2", "refbench"], default="config2",
                    help="config2 = the metric's workload (default).  refbench = the shape of the reference's own published "
                         "benchmark (benchmarks/jwas_nonblock_benchmark.jl:34-51): X ~ U[0,1) fp32 uncentred, y ~ N(0,1), Pi = 0 "
                         "(every marker in the model, estimatePi = false), marker variance fixed; use with --p 100000 / 200000")
    ap.add_argument("--storage", choices=["dense", "packed2bit"], default="dense",
                    help="dense = the metric's fp32 dense genotypes (default); packed2bit = the reference's 2-bit packed "
                         "streaming payload kept packed in HBM (same genotypes, same chain; extra, not the headline config)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample-markers", type=int, default=4000)
    return ap.parse_args()


_T0 = time.time()


def log(msg):
    if os.environ.get("JWAS_BENCH_VERBOSE", "0") != "0" and int(os.environ.get("RANK", "0")) == 0:
        print(f"[bench {time.time() - _T0:8.2f}s] {msg}", file=sys.stderr, flush=True)


def main():
    a = parse()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    import torch
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        # JWAS_BENCH_BACKEND=gloo + JWAS_BENCH_ONE_DEVICE=1: debug only -- exercises the N > 1 code path on a box with
        # a single GPU (all ranks share device 0, the exchange goes through host memory)
        backend = os.environ.get("JWAS_BENCH_BACKEND", "nccl")
        if os.environ.get("JWAS_BENCH_ONE_DEVICE"):
            local_rank = 0
        torch.cuda.set_device(local_rank)
        dist.init_process_group(backend, rank=rank, world_size=world)
    if world != a.gpus:
        if rank == 0:
            print(f"warning: --gpus {a.gpus} but WORLD_SIZE={world}; using WORLD_SIZE", file=sys.stderr)
    import jwas_jl_amd as J
    from jwas_jl_amd.dist import MarkerShard, shard_range

    from jwas_jl_amd.mcmc import pick_block_size
    refbench = a.workload == "refbench"
    adaptive = a.block_size == 0 and not refbench
    n, p_total, bs = a.n, a.p, (512 if adaptive else (a.block_size or 128))      # dense prior (refbench): 128-marker blocks
    lo, hi = shard_range(p_total, rank, world, align=1024 if adaptive else bs)
    p_loc = hi - lo
    eng = J.HipEngine(local_rank)
    t_setup = time.time()
    log('engine created'); (eng.alloc_packed if a.storage == 'packed2bit' else eng.alloc_dense)(n, p_loc); log('alloc done')
    if refbench:
        eng.synth(a.seed, kind=1, center=False, marker_offset=lo)   # X ~ U[0,1), uncentred (jwas_nonblock_benchmark.jl:38,46)
    else:
        eng.synth(a.seed, kind=0, center=True, marker_offset=lo)    # 0/1/2 genotypes, centred, generated on device
    log('synth done'); eng.setup_blocks(bs, "mfma")
    if adaptive:
        eng.add_block_size(1024, "mfma")
    log('setup_blocks done')
    eng.init_state("BayesC", 1)
    shard = MarkerShard(eng, lo, hi, rank, world)

    # ---- simulate y = 1 + X beta + e with ncausal QTL, h2 = 0.5
    rng = np.random.default_rng(a.seed)
    ncausal = max(1, p_total // 1000)
    causal = np.sort(rng.choice(p_total, size=ncausal, replace=False))
    eff = rng.standard_normal(ncausal)
    a_true = np.zeros(p_loc, dtype=np.float32)
    m = (causal >= lo) & (causal < hi)
    a_true[causal[m] - lo] = eff[m]
    eng.set_state(alpha=a_true)
    g = shard.allreduce_sum(eng.mul_alpha().astype(np.float64)); log('mul_alpha done')
    g *= np.sqrt(0.5 / g.var())
    y = (1.0 + g + rng.standard_normal(n) * np.sqrt(0.5)).astype(np.float32)
    if refbench:
        y = rng.standard_normal(n).astype(np.float32)                # y1 = randn(Float32, n)  (:35)
    eng.set_state(alpha=np.zeros(p_loc), beta=np.zeros(p_loc), delta=np.ones(p_loc))

    # ---- priors (input_data_validation.jl:296-350, tools4genotypes.jl:353-478)
    df_ = 4.0
    vary = float(np.var(y.astype(np.float64), ddof=1))
    vare = np.float32(0.5 * vary)
    sum2pq = float(shard.allreduce_sum(np.array([eng.xpx().astype(np.float64).sum()]))[0]) / n   # x'x/n = 2pq (centred)
    pi = 0.95
    Gval = np.float32(0.5 * vary / ((1.0 - pi) * sum2pq))
    if refbench:      # get_genotypes(X, 1.0; ...) and build_model(..., 1.0): genetic variance 1, residual variance 1, Pi = 0
        xbar2 = 0.25                                                 # alleleFreq = mean/2 of U[0,1) columns (readgenotypes.jl:385)
        pi, vare = 0.0, np.float32(1.0)
        Gval = np.float32(1.0 / (p_total * 2.0 * xbar2 * (1.0 - xbar2)))
    scale_e = float(vare) * (df_ - 2) / df_
    scale_g = float(Gval) * (df_ - 2) / df_
    setup_s = time.time() - t_setup; log(f'setup done {setup_s:.1f}s')

    state = {"r": y[None, :].copy(), "mu": 0.0, "vare": vare, "G": Gval, "pi": pi, "it": 0, "bs": bs}
    acc = {"sweep_ms": 0.0, "k_ms": 0.0, "k_n": 0.0, "k_bytes": 0.0, "events": 0.0, "ovh_ms": 0.0, "launches": 0.0, "bytes": 0.0}

    def step():
        s = state
        s["it"] += 1
        # 1. intercept: single-site Gibbs on the 1x1 MME (solver.jl:143-151)
        r = s["r"][0].astype(np.float64) + s["mu"]
        s["mu"] = rng.standard_normal() * np.sqrt(float(s["vare"]) / n) + r.sum() / n
        r -= s["mu"]
        # 2. marker sweep on the device (+ shard reconcile)
        r_new, st = shard.sweep(r.astype(np.float32)[None, :], iteration=s["it"], seed=a.seed,
                                vare=s["vare"], var_effect=s["G"], pi=s["pi"], nreps=1)
        s["r"] = r_new
        if adaptive:       # n_events is the all-shard total after the reconcile: every rank takes the same decision
            eng.select_block_size(pick_block_size(st["n_events"], p_total))
        acc["launches"] += -(-p_loc // s["bs"]) + 1
        acc["bytes"] += 4.0 * n * p_loc if a.storage == "dense" else 0.25 * n * p_loc
        s["bs"] = eng.block_size
        nl = st["sum_delta"][0]
        # 3-5. pi, marker-effect variance, residual variance (Pi.jl:7-9, variance_components.jl:60-66,151-162)
        if not refbench:                                             # refbench: estimatePi = false, estimate_variance = false
            s["pi"] = float(rng.beta(p_total - nl + 1.0, nl + 1.0))
            s["G"] = np.float32((np.float32(st["alpha_ss"][0, 0]) + df_ * scale_g) / rng.chisquare(nl + df_))
        s["vare"] = np.float32((np.float32(st["resid_ss"][0, 0]) + df_ * scale_e) / rng.chisquare(n + df_))
        acc["sweep_ms"] += st["sweep_ms"]
        acc["k_ms"] += st["update_kernel_ms"]
        acc["k_n"] += st["update_kernel_samples"]
        acc["k_bytes"] += st["update_kernel_bytes"]
        acc["events"] += st["n_events"]
        acc["ovh_ms"] = st["event_overhead_ms"]
        return st

    def barrier():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    # chain burn-in (setup): only when the warm-up alone would not reach the steady state
    nburn = max(0, a.burnin - a.warmup)
    for _ in range(nburn):
        step()
    log(f"burn-in done: {nburn} sweeps")
    for _ in range(a.warmup):
        st_ = step(); log(f"warmup step: sweep_ms={st_['sweep_ms']:.1f} events={st_['n_events']:.0f} in_model={st_['sum_delta'][0]:.0f}")
    for k in acc:
        acc[k] = 0.0
    barrier()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        last = step()
    barrier()
    elapsed = time.perf_counter() - t0
    log(f'timed region done: {elapsed:.2f}s')
    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64, device="cuda" if torch.distributed.get_backend() == "nccl" else "cpu")
        torch.distributed.all_reduce(tt, op=torch.distributed.ReduceOp.MAX)
        elapsed = float(tt.item())

    out = None
    if rank == 0:
        ms_per_step = 1e3 * elapsed / a.steps
        # Dominant kernel: k_block_step, one launch per marker block.  Its average launch duration is taken from
        # the HIP events the library records on the sweep's stream around each sweep (start of the first step,
        # end of the last): sweep time / launches.  (Events around individual launches cost ~3 us each and would
        # perturb the timed region; the inter-launch gap is therefore included -- a conservative duration.
        # The rocprofv3 --kernel-trace --stats average of the same command is committed under profiles/.)
        launches = acc["launches"]                                 # k_block_step launches of the timed sweeps (nblocks + 1 each)
        avg_launch_us = 1e3 * acc["sweep_ms"] / launches
        bytes_per_launch = acc["bytes"] / launches                 # algorithmic: 4 B (2 bits if packed) x n per marker, X read once
        bs = state["bs"]
        achieved = bytes_per_launch / 1e9 / (avg_launch_us * 1e-6)
        out = {
            "metric": "MCMC iters/sec (full marker sweep)", "value": a.steps / elapsed, "unit": "iterations/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": ms_per_step,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": (f"reference benchmark shape (jwas_nonblock_benchmark.jl): BayesC, {n} x {p_total}, X~U[0,1) fp32 uncentred, y~N(0,1), Pi=0 fixed, marker variance fixed"
                                    if refbench else
                                    f"single-trait BayesC, {n} individuals x {p_total} SNPs, " + ("2-bit packed genotypes (decoded to fp32 on the fly)" if a.storage == "packed2bit" else "fp32 dense genotypes") + ", pi0=0.95 estimated"),
                       "storage": a.storage,
                       "n": n, "p": p_total, "block_size": bs, "block_policy": "adaptive 512/1024" if adaptive else "fixed", "parallelism": f"marker-shard x{world}" if world > 1 else "single GPU",
                       "device_sweep_ms": acc["sweep_ms"] / a.steps, "events_per_sweep": acc["events"] / a.steps,
                       "markers_in_model": float(last["sum_delta"][0]), "setup_s": setup_s,
                       "chain_sweeps_before_timing": nburn + a.warmup},
            "roofline": {"bound": "hbm", "kernel": "k_block_step", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": TRAFFIC_BYTES_PER_LAUNCH if (TRAFFIC_BYTES_PER_LAUNCH and adaptive and bs == 1024 and p_total == P_TOTAL and n == N_IND and world == 1 and a.storage == "dense") else None,
                         "bytes_per_launch": bytes_per_launch, "avg_launch_us": avg_launch_us, "launches_timed": launches},
        }
        if world == 1 and not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(eng, n, p_total, min(a.cpu_sample_markers, p_loc), y, float(vare), float(Gval))
    eng.close()
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()
    if rank == 0:
        print(json.dumps(out))


def cpu_baseline(eng, n, p_total, p_sub, y, vare, Gval):
    """The CPU oracle's non-block BayesC sweep (oracle/jwas_oracle.c: orc_time_bayesc_sweeps -- per marker
    fp32 dot, scalar update, conditional fp32 axpy: the reference's operation order, BayesABC.jl:60-80) on
    the first p_sub markers of the same matrix, scaled linearly in p (the reference's own projection
    device, benchmarks/streaming_large_benchmark.jl:161-184).  Timed single-threaded and with the
    dot/axpy rows split over a few threads (what a threaded BLAS does per marker; more threads than ~8
    only add fork/join cost on 50k-element vectors); the faster one is reported with its thread count."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle as O
    X = eng.get_columns(0, p_sub)
    xpx = O.xpx(X, O.ACC_F32)
    ncores = os.cpu_count() or 1
    res = {}
    for threads in sorted({1, min(ncores, 8)}):
        r = (y - y.mean()).astype(np.float32)
        al = np.zeros(p_sub, dtype=np.float32)
        be = np.zeros(p_sub, dtype=np.float32)
        de = np.zeros(p_sub, dtype=np.float32)
        # one calibration sweep on a tenth of the sample, then ~6 s of timed work
        pc = max(200, p_sub // 10)
        t1 = O.time_bayesc_sweeps(np.asfortranarray(X[:, :pc]), xpx[:pc].copy(), r.copy(), al[:pc].copy(), be[:pc].copy(),
                                  de[:pc].copy(), vare, Gval, 0.95, 1, 1, threads) * (p_sub / pc)
        sweeps = int(max(1, min(10, round(6.0 / max(t1, 1e-3)))))
        tt = O.time_bayesc_sweeps(X, xpx, r, al, be, de, vare, Gval, 0.95, 1, sweeps, threads)
        res[threads] = (tt / sweeps, sweeps)
        log(f"cpu baseline {threads} thread(s): {tt / sweeps:.3f} s per {p_sub}-marker sweep")
    best = min(res, key=lambda k: res[k][0])
    per_sweep_full = res[best][0] * p_total / p_sub
    detail = "; ".join(f"{k} thread(s): {v[0] * p_total / p_sub:.1f} s/sweep" for k, v in sorted(res.items()))
    return {"value": 1.0 / per_sweep_full, "unit": "iterations/s", "cores": best, "kind": "port",
            "sample": (f"{res[best][1]} sweeps over the first {p_sub} of {p_total} markers (n={n}), scaled linearly in p; "
                       f"{detail}; host has {ncores} logical cores; sweep only, host updates excluded")}


if __name__ == "__main__":
    main()
