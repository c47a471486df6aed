#!/usr/bin/env python
"""bench.py -- MCMC iterations/sec of the marker-effect Gibbs sampler on MI355X.

Metric (BASELINE.json): MCMC iters/sec (full marker sweep), 50k x 600k single-trait BayesC, fp32 dense
genotypes, pi0 = 0.95 estimated, at 1/2/4/8 GPUs.  One "step" = one MCMC iteration = one full sweep
over all p markers (device) + the host-side updates of MCMC_BayesianAlphabet.jl:196-220,294-370
(location parameters, pi, marker-effect variance, residual variance).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload ...]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Workloads (BASELINE.json `configs`):
    config2       single-trait BayesC, 50 000 x 600 000 fp32 dense, pi0 = 0.95 estimated  (the metric; default)
                  --pi-fixed 0.95: estimatePi = false -- the high-turnover variant (~30 000 markers move per sweep)
    config3       single-trait BayesR (gamma 0, .01, .1, 1; pi0 = .95, .03, .015, .005 estimated), 50 000 x 600 000
    config4       3-trait BayesC, Gibbs sampler I, 20 000 x 100 000, R and G inverse-Wishart on the host, 8-state pi table
                  estimated; --mt-prior default (the reference's default: all mass on the all-ones state, every marker
                  in the model) | sparse (0.95 on the null state)
    config5shard  one GPU's share of config 5 (single-step shaped input): BayesC, 280 000 rows (80 000 integer-coded +
                  200 000 real-valued "imputed") x 75 000 markers = 84 GB; with --gpus N every rank holds its own 75 000
                  markers (N = 8 is config 5 itself) -- weak scaling
    refbench      the shape of the reference's own published benchmark (benchmarks/jwas_nonblock_benchmark.jl:34-51)

N > 1 (config2/3/4: total work fixed, "strong"): one process per GPU.  Under torch.distributed.run (RANK / WORLD_SIZE in
the environment) this process is one rank; run plainly with --gpus N > 1 it SPAWNS the N ranks itself (re-executes
under torch.distributed.run on 127.0.0.1) and fails -- exit code 2, no JSON line -- when the box has fewer GPUs than N.
  --shard markers (default): markers are sharded over the ranks (jwas.jl_amd/dist.py MarkerShard): each rank sweeps its markers
                  from the same residual snapshot and ONE ncclAllReduce of the residual delta + the packed marker
                  statistics reconciles per sweep, inside the library (jwas_hip_sweep_sharded), residual resident in HBM;
  --shard rows:   exact row shards (dist.RowShard / jwas_hip_comm_row_shards): every rank holds n / N individuals and all
                  markers, one small all-reduce of the block's partial right-hand side per block launch, sampler replicated.
The line's n_gpus is the rank count the RCCL communicator itself reports (ncclCommCount); a mismatch with --gpus is an error.

Prints ONE JSON line (rank 0) with the contract fields plus
  "roofline":     HBM roofline of the dominant kernel (k_block_step: sampler of block k-1 || update + partial
                  RHS of block k): algorithmic bytes per launch / average launch duration, from the HIP events
                  recorded on the sweep's stream inside the timed region; "traffic" is filled from the PMC summaries
                  under profiles/ when they were collected for exactly this configuration (traffic_source), else null;
  "cpu_baseline": the CPU oracle's non-block sweep of the same sampler (the reference's per-marker sdot / scalar update /
                  saxpy order) timed on this box's host cores on a marker subsample (N = 1, rank 0 only): single thread,
                  a 16-thread team and all cores; the fastest is `value`, all are listed in `sample`;
  "via_api":      (config2, N = 1) the same chain driven through the package's runMCMC() -- the API path a user calls.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0                       # MI355X_MICROARCH.md: HBM3E 8 TB/s peak
BAYESR_GAMMA = np.array([0.0, 0.01, 0.1, 1.0])

WORKLOADS = {
    #                n        p        method     traits
    "config2":      (50_000,  600_000, "BayesC",  1),
    "config3":      (50_000,  600_000, "BayesR",  1),
    "config4":      (20_000,  100_000, "MTBayesC", 3),
    "config5shard": (280_000, 75_000,  "BayesC",  1),
    "refbench":     (50_000,  100_000, "BayesC",  1),
}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=30)
    ap.add_argument("--burnin", type=int, default=30,
                    help="chain iterations run as part of the SETUP (untimed, before the warm-up steps): the metric is defined "
                         "on the steady state of the chain (SURVEY.md section 8d), which a start from alpha = 0 reaches after "
                         "~25 sweeps; with --warmup >= 30 (the default) nothing extra runs")
    ap.add_argument("--chain", type=int, default=0,
                    help="run this many chain sweeps FROM THE START before the warm-up (instead of --burnin) and record every one of them: "
                         "the line gets a `chain` object -- chain_total_s, worst sweep, means per 100-sweep window with the effect changes, "
                         "markers in the model and block size of each window -- so that a workload whose chain passes through several "
                         "regimes (config 3: BayesR sheds markers for hundreds of sweeps; config 4: dense start -> transition -> sparse) is "
                         "reported on its transient AND on its steady state (the timed region that follows)")
    ap.add_argument("--workload", choices=sorted(WORKLOADS), default="config2")
    ap.add_argument("--n", type=int, default=0, help="override the workload's number of individuals")
    ap.add_argument("--p", type=int, default=0, help="override the workload's number of markers (per GPU for config5shard)")
    ap.add_argument("--block-size", type=int, default=int(os.environ.get("JWAS_BLOCK_SIZE", "0")),
                    help="0 (default) = the host loop's policy (jwas.jl_amd/mcmc.py); or a fixed size in {64,...,1024}")
    ap.add_argument("--pi-fixed", type=float, default=None,
                    help="BayesC workloads: keep pi at this value (estimatePi = false) -- the high-turnover variant")
    ap.add_argument("--mt-prior", choices=["default", "sparse"], default="default")
    ap.add_argument("--no-section-solve", action="store_true",
                    help="dense priors: the sequential walk instead of Rule T (jwas_sweep_params.section_solve) -- A/B")
    ap.add_argument("--mt-method", choices=["BayesC", "BayesB"], default="BayesC",
                    help="config4: BayesB = multi-trait BayesA/B, one effect covariance per marker (redrawn on the host each iteration)")
    ap.add_argument("--shard", choices=["markers", "rows"], default="markers",
                    help="N > 1: markers = marker shards + one all-reduce of the residual delta per sweep (the metric's mode); "
                         "rows = exact row shards (one small all-reduce per block launch, replicated sampler)")
    ap.add_argument("--one-rank-comm", action="store_true",
                    help="with --gpus 1: run the sweep through the library's sharded path (jwas_hip_sweep_sharded: snapshot, pack "
                         "kernel, ncclAllReduce on a ONE-rank RCCL communicator, apply kernel) -- what one rank of an N-GPU job "
                         "executes per iteration, minus the wire time (profiles/r04_rank_share.json)")
    ap.add_argument("--seed", type=int, default=2026)
    ap.add_argument("--storage", choices=["dense", "packed2bit"], default="dense",
                    help="dense = the metric's fp32 dense genotypes (default); packed2bit = the reference's 2-bit packed "
                         "streaming payload kept packed in HBM (same genotypes, same chain; extra, not the headline config)")
    ap.add_argument("--groups", type=int, default=int(os.environ.get("JWAS_BENCH_GROUPS", "-1")), choices=[-1, 0, 2, 4],
                    help="blocks per launch of the step kernel in the sparse steady state (grouped launches, jwas_hip_setup_groups); "
                         "0 = one block per launch; -1 = the host policy's default (mcmc.GROUPED_BLOCKS_PER_LAUNCH)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample-markers", type=int, default=20000)
    ap.add_argument("--cpu-seconds", type=float, default=8.0, help="target CPU time per thread-count leg")
    ap.add_argument("--via-api", type=int, default=-1,
                    help="timed iterations of the runMCMC() leg (config2, 1 GPU; default 20, 0 = skip)")
    if "JWAS_BENCH_ARGV" in os.environ and "RANK" in os.environ and len(sys.argv) == 1:      # a rank started by spawn_ranks()
        return ap.parse_args(json.loads(os.environ["JWAS_BENCH_ARGV"]))
    return ap.parse_args()


_T0 = time.time()


def log(msg):
    if os.environ.get("JWAS_BENCH_VERBOSE", "0") != "0" and int(os.environ.get("RANK", "0")) == 0:
        print(f"[bench {time.time() - _T0:8.2f}s] {msg}", file=sys.stderr, flush=True)


def traffic_from_profiles(workload, n, p, bs, storage, world, pi_fixed, variant=None, groups=0):
    """HBM bytes per k_block_step launch from the PMC summaries committed under profiles/ (rocprofv3 --pmc FETCH_SIZE and
    --pmc WRITE_SIZE in separate runs of this command; FETCH_SIZE x2 per the guide's gfx950 correction, checked on k_xpx
    which reads X exactly once).  Only returned when the summaries' recorded configuration matches this run."""
    path = os.path.join(ROOT, "profiles", "traffic.json")
    if not os.path.exists(path):
        return None, None
    try:
        with open(path) as fh:
            rows = json.load(fh)
    except (OSError, ValueError):
        return None, None
    for row in rows:
        c = row.get("config", {})
        if (c.get("workload") == workload and c.get("n") == n and c.get("p") == p and c.get("block_size") == bs
                and c.get("storage") == storage and c.get("n_gpus") == world and c.get("pi_fixed") == pi_fixed
                and c.get("variant") == variant and int(c.get("blocks_per_launch", 0) or 0) == (groups if groups >= 2 else 0)):
            return float(row["bytes_per_launch"]), row.get("source")
    return None, None


def spawn_ranks(a):
    """`python bench.py --gpus N` run plainly (no RANK in the environment): become the launcher of N ranks, one per GPU,
    the same way the driver's torch.distributed.run command line does.  Never measures fewer GPUs than asked for."""
    import torch
    ndev = torch.cuda.device_count()
    if ndev < a.gpus and not os.environ.get("JWAS_BENCH_ONE_DEVICE"):
        print(f"bench.py: --gpus {a.gpus} needs {a.gpus} GPUs, this box has {ndev}; refusing to report a {ndev}-GPU number "
              f"as a {a.gpus}-GPU one", file=sys.stderr)
        sys.exit(2)
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 1) // max(1, a.gpus))))
    # the ranks read this process's own argument list from the environment: torch.distributed.run's parser would try to
    # match options placed after the script name against its own abbreviations (--n ...)
    env["JWAS_BENCH_ARGV"] = json.dumps(sys.argv[1:])
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)]
    sys.exit(subprocess.call(cmd, env=env))


def main():
    a = parse()
    if "RANK" not in os.environ and a.gpus > 1:
        spawn_ranks(a)
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        if rank == 0:
            print(f"bench.py: --gpus {a.gpus} but the launcher started {world} rank(s); refusing to print a line whose n_gpus "
                  "is not the number of ranks", file=sys.stderr)
        sys.exit(2)
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
        elif torch.cuda.device_count() <= local_rank:
            print(f"bench.py: rank {rank} has no GPU {local_rank} (the box has {torch.cuda.device_count()})", file=sys.stderr)
            sys.exit(2)
        torch.cuda.set_device(local_rank)
        dist.init_process_group(backend, rank=rank, world_size=world)
    elif a.one_rank_comm:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.cuda.set_device(0)
        dist.init_process_group("nccl", rank=0, world_size=1)
    import jwas_jl_amd as J
    from jwas_jl_amd.dist import MarkerShard, RowShard, shard_range
    from jwas_jl_amd.mcmc import pick_block_size, pick_block_size_mt, mt_1024_allowed, MT_1024_CHANGE_FRACTION

    wl = a.workload
    n, p_arg, method, t = WORKLOADS[wl]
    n = a.n or n
    p_arg = a.p or p_arg
    weak = wl == "config5shard"
    p_total = p_arg * world if weak else p_arg
    refbench = wl == "refbench"
    bayesc = method == "BayesC"
    estimate_pi = not refbench and a.pi_fixed is None
    mt_dense = (t > 1 and a.mt_prior == "default")
    dense_prior = refbench or mt_dense or (a.pi_fixed is not None and a.pi_fixed < 0.5)
    # block policy = mcmc.run_chain's: dense priors 128; sparse single-trait priors adaptive 512/1024; multi-trait 512
    adaptive = a.block_size == 0 and not dense_prior and t == 1
    # (single-trait priors that include every marker whatever its rhs -- refbench's Pi = 0 -- run 512-marker blocks through
    # the sampler's dense_big_st path; the multi-trait default prior keeps 128)
    all_in = refbench or (t == 1 and bayesc and a.pi_fixed == 0.0)
    mt_big = mt_dense                                   # (sampler I, shared or per-marker covariance: 256-marker blocks through dense_big_mt)
    bs = a.block_size or ((512 if all_in else (256 if mt_big else 128)) if dense_prior else 512)
    # Rule T (mcmc.run_chain's policy): the dense 256-marker blocks of sampler I as triangular solves; --no-section-solve = the walk
    section_solve = bool(mt_big and t <= 3 and bs == 256 and not a.no_section_solve)
    adaptive_mt = bool(a.block_size == 0 and mt_big and bs == 256 and 512 * t <= 2048)      # (256 while the chain is dense, 512 once it is sparse,
    mt_1024 = adaptive_mt and mt_1024_allowed(t, p_total, a.mt_method == "BayesB")           #  1024 once a block holds a handful of candidates)
    # (a multi-trait chain that starts sparse: 512-marker blocks, 1024 below 0.5 % turnover -- mcmc.run_chain's policy)
    adaptive_mts = bool(a.block_size == 0 and t > 1 and not dense_prior and bs == 512 and mt_1024_allowed(t, p_total, a.mt_method == "BayesB"))
    rows_mode = a.shard == "rows"
    if rows_mode and (weak or a.storage != "dense"):
        raise SystemExit("--shard rows runs the dense strong-scaling workloads (config2 / config3 / config4 / refbench)")
    if rows_mode:
        # exact row shards: every rank holds n / N individuals and ALL markers (the same number of 256-row groups on every rank)
        lo, hi = 0, p_total
        n_loc = -(-n // world)
        n_loc = min(n_loc, n - rank * n_loc) if rank == world - 1 else n_loc
        if -(-n_loc // 256) != -(-(-(-n // world)) // 256):
            raise SystemExit(f"--shard rows: {n} individuals do not split into {world} slices with the same number of 256-row groups")
        synth_seed = a.seed + 1_000_003 * rank      # every rank generates its OWN individuals (centred within the slice: the pooled columns sum to zero too)
    else:
        lo, hi = shard_range(p_total, rank, world, align=1024 if adaptive else bs)
        n_loc, synth_seed = n, a.seed
    p_loc = hi - lo
    eng = J.HipEngine(local_rank)
    t_setup = time.time()
    log('engine created'); (eng.alloc_packed if a.storage == 'packed2bit' else eng.alloc_dense)(n_loc, p_loc); log('alloc done')
    n_gen = 0
    if refbench:
        eng.synth(synth_seed, kind=1, center=False, marker_offset=lo)   # X ~ U[0,1), uncentred (jwas_nonblock_benchmark.jl:38,46)
    elif weak:
        n_gen = int(round(n * 80_000 / 280_000))                    # 80k genotyped of 280k phenotyped rows (config 5)
        eng.synth_single_step(a.seed, n_gen, center=True, marker_offset=lo)
    else:
        eng.synth(synth_seed, kind=0, center=True, marker_offset=lo)    # 0/1/2 genotypes, centred, generated on device
    log('synth done')
    mt_pervar = t > 1 and a.mt_method == "BayesB"
    if mt_pervar and world > 1:
        raise SystemExit("--mt-method BayesB runs on one GPU")
    if rows_mode:
        shard = RowShard(eng, rank, world)           # before setup_blocks: x'x and the Grams are summed over the ranks there
    eng.setup_blocks(bs, "mfma")
    if adaptive:
        eng.add_block_size(1024, "mfma")
    elif adaptive_mt:
        eng.add_block_size(512, "mfma")
        if mt_1024:
            eng.add_block_size(1024, "mfma")
    elif adaptive_mts:
        eng.add_block_size(1024, "mfma")
    # grouped launches (single-trait sparse steady state: the 1024-marker set of the adaptive policy, or the fixed block size of a
    # packed / explicitly sized run).  The metric is defined on the steady state of a long chain, so the bench sets the
    # groups up whatever --steps is; mcmc.run_chain asks for them only when the chain is long enough to pay for the set-up
    # (mcmc.grouped_blocks_for_chain) -- the set-up cost is reported in config.group_setup_s, outside the timed region like every Gram.
    from jwas_jl_amd.mcmc import GROUPED_BLOCKS_PER_LAUNCH, grouped_launch_size
    groups = GROUPED_BLOCKS_PER_LAUNCH if a.groups < 0 else a.groups
    group_setup_s = 0.0
    group_bs = grouped_launch_size(method, t, rows_mode, (1024 if adaptive else bs), groups, dense_prior=dense_prior)
    if group_bs:
        cur = eng.block_size
        eng.select_block_size(group_bs)
        try:
            t_g = time.time()
            eng.setup_groups(groups, "mfma")
            group_setup_s = time.time() - t_g
        except Exception as ex:      # noqa: BLE001  (e.g. no HBM left for the group cross-Grams: one block per launch, said in the log and in config.blocks_per_launch)
            log(f"grouped launches not set up ({ex}); running one block per launch")
        eng.select_block_size(cur)
    # ... and ping-pong pairs on the 512-marker set of a high-turnover chain (BayesR, a fixed pi): mcmc.pingpong_pairs_for_chain
    from jwas_jl_amd.mcmc import pingpong_pairs_for_chain
    pair_m = int(os.environ["JWAS_BENCH_GROUPS_SMALL"]) if os.environ.get("JWAS_BENCH_GROUPS_SMALL") else \
        (pingpong_pairs_for_chain(method, estimate_pi, 10 ** 6) if (groups and a.storage == "dense") else 0)
    pairs_on = False
    if pair_m and adaptive and not rows_mode and grouped_launch_size(method, t, rows_mode, 512, pair_m, dense_prior=dense_prior):
        try:
            t_g = time.time()
            eng.setup_groups(pair_m, "mfma")          # (the selected size is the 512-marker set here)
            group_setup_s += time.time() - t_g
            pairs_on = True
        except Exception as ex:      # noqa: BLE001
            log(f"ping-pong pairs not set up ({ex})")
    log('setup_blocks done')
    eng.init_state("MTBayesB" if mt_pervar else method, t)
    if not rows_mode:
        shard = MarkerShard(eng, lo, hi, rank, world, force_collective=a.one_rank_comm)
    comm_world = shard.comm_world()
    if comm_world != world:
        raise SystemExit(f"bench.py: the communicator reports {comm_world} rank(s), the launcher started {world}")

    # ---- simulate y_k = 1 + X beta_k + e_k with ncausal QTL, h2 = 0.5
    rng = np.random.default_rng(a.seed)
    rng_e = np.random.default_rng(a.seed + 17 + (rank if rows_mode else 0))      # residuals: per individual (rows mode: per rank)
    ncausal = max(1, p_total // 1000)
    causal = np.sort(rng.choice(p_total, size=ncausal, replace=False))
    m = (causal >= lo) & (causal < hi)
    Y = np.empty((t, n_loc), dtype=np.float32)
    for k in range(t):
        eff = rng.standard_normal(ncausal)
        a_true = np.zeros(p_loc, dtype=np.float32)
        a_true[causal[m] - lo] = eff[m]
        eng.set_state(0, alpha=a_true)
        g = eng.mul_alpha(0).astype(np.float64)
        if rows_mode:       # own individuals, all markers: pooled variance of g over the ranks
            mom = shard.allreduce_sum(np.array([g.sum(), (g * g).sum()]))
            gvar = mom[1] / n - (mom[0] / n) ** 2
        else:               # own markers, all individuals
            g = shard.allreduce_sum(g)
            gvar = g.var()
        g *= np.sqrt(0.5 / gvar)
        Y[k] = (1.0 + g + rng_e.standard_normal(n_loc) * np.sqrt(0.5)).astype(np.float32)
        if refbench:
            Y[k] = rng_e.standard_normal(n_loc).astype(np.float32)     # y1 = randn(Float32, n)  (:35)
    log('phenotypes done')
    dlt0 = np.ones(p_loc, dtype=np.int32 if method == "BayesR" else np.float32)
    for k in range(t):
        eng.set_state(k, alpha=np.zeros(p_loc), beta=np.zeros(p_loc), delta=dlt0)

    # ---- priors (input_data_validation.jl:296-350, tools4genotypes.jl:353-478, build_MME.jl:128-141)
    Y64 = Y.astype(np.float64)
    if rows_mode:
        mom = shard.allreduce_sum(np.concatenate([Y64.sum(axis=1), (Y64 * Y64).sum(axis=1)]))
        vary = (mom[t:] - mom[:t] ** 2 / n) / (n - 1)
        sum2pq = float(eng.xpx().astype(np.float64).sum()) / n                      # (x'x is already the pooled one)
    else:
        vary = np.array([float(np.var(Y64[k], ddof=1)) for k in range(t)])
        sum2pq = float(shard.allreduce_sum(np.array([eng.xpx().astype(np.float64).sum()]))[0]) / n   # x'x/n = 2pq (centred)
    nstates = 1 << t
    if t == 1:
        df_e = df_g = 4.0
        vare = np.float32(0.5 * vary[0])
        if method == "BayesR":
            pi = np.array([0.95, 0.03, 0.015, 0.005])
            Gval = np.float32(0.5 * vary[0] / (sum2pq * float((BAYESR_GAMMA * pi).sum())))
        else:
            pi = 0.95 if a.pi_fixed is None else float(a.pi_fixed)
            Gval = np.float32(0.5 * vary[0] / ((1.0 - pi) * sum2pq))
        if refbench:      # get_genotypes(X, 1.0; ...) and build_model(..., 1.0): genetic variance 1, residual variance 1, Pi = 0
            xbar2 = 0.25                                             # alleleFreq = mean/2 of U[0,1) columns (readgenotypes.jl:385)
            pi, vare = 0.0, np.float32(1.0)
            Gval = np.float32(1.0 / (p_total * 2.0 * xbar2 * (1.0 - xbar2)))
        scale_e = float(vare) * (df_e - 2) / df_e
        scale_g = float(Gval) * (df_g - 2) / df_g
    else:
        df_e = df_g = 4.0 + t                                        # build_MME.jl:108-110,131-134
        vare = np.diag(0.5 * vary).astype(np.float32)
        pi = np.zeros(nstates)
        if a.mt_prior == "default":
            pi[nstates - 1] = 1.0                                    # tools4genotypes.jl:357-373
        else:
            pi[0] = 0.95
            pi[1:] = 0.05 / (nstates - 1)
        Gval = (np.diag(0.5 * vary) / (sum2pq * pi[nstates - 1])).astype(np.float32)    # :426-438
        scale_e = np.asarray(vare, dtype=np.float64) * (df_e - t - 1)
        scale_g = np.asarray(Gval, dtype=np.float64) * (df_g - t - 1)
    setup_s = time.time() - t_setup; log(f'setup done {setup_s:.1f}s')

    # The residual lives on the device for the whole chain: the host's location step is the intercept, whose Gibbs update
    # needs sum(r) (returned with every sweep's statistics) and whose residual correction is a scalar shift
    # (jwas_hip_residual_add_scalar).  No O(n) host traffic per iteration.
    for k in range(t):
        eng.set_residual(Y[k], k)
    rsum0 = Y64.sum(axis=1)
    if rows_mode:
        rsum0 = shard.allreduce_sum(rsum0)
    state = {"rsum": rsum0, "mu": np.zeros(t), "vare": vare, "G": Gval, "pi": pi, "it": 0, "bs": bs}
    if t > 1 and a.mt_method == "BayesB":
        state["Gmat"] = np.tile(np.asarray(Gval, dtype=np.float32), (p_loc, 1, 1))      # MCMC_BayesianAlphabet.jl:67-69
    acc = {"sweep_ms": 0.0, "events": 0.0, "launches": 0.0, "bytes": 0.0}
    from jwas_jl_amd.engine import SectionSolvePolicy
    solve_policy = SectionSolvePolicy(section_solve, 4 * (p_loc // 256))       # (mcmc.run_chain's: off while most sections are not solved)

    def step():
        s = state
        s["it"] += 1
        # 1. intercepts: single-site Gibbs on the MME (solver.jl:143-162); sum(r adjusted for mu) = sum(r) + n mu
        mu_old = s["mu"].copy()
        rs = s["rsum"] + n * mu_old
        if t == 1:
            s["mu"][0] = rng.standard_normal() * np.sqrt(float(s["vare"]) / n) + rs[0] / n
        else:
            Rinv = np.linalg.inv(np.asarray(s["vare"], dtype=np.float64))
            A, b = n * Rinv, Rinv @ rs
            for k in range(t):
                il = 1.0 / A[k, k]
                s["mu"][k] = rng.standard_normal() * np.sqrt(il) + il * (b[k] - A[:, k] @ s["mu"]) + s["mu"][k]
        for k in range(t):
            eng.residual_add_scalar(mu_old[k] - s["mu"][k], k)
        # 2. marker sweep on the device (+ shard reconcile)
        kw = dict(iteration=s["it"], seed=a.seed, vare=s["vare"], var_effect=s["G"], nreps=1)
        if solve_policy.use(s["it"]) and eng.block_size == 256:
            kw["section_solve"] = True
        if method == "BayesR":
            kw["pi_classes"] = s["pi"]
        elif t > 1:
            with np.errstate(divide="ignore"):
                kw["log_prior_states"] = np.log(s["pi"])
            if mt_pervar:
                kw["var_effect_matrix"] = s["Gmat"]
        else:
            kw["pi"] = s["pi"]
        m_now = eng.blocks_per_launch()
        if m_now >= 2:
            kw["group_launch"] = True
        st = shard.sweep_resident(**kw)
        s["rsum"] = np.asarray(st["resid_sum"], dtype=np.float64).copy()
        solve_policy.observe(s["it"], eng, ran=bool(kw.get("section_solve")))
        if adaptive:       # n_events is the all-shard total after the reconcile: every rank takes the same decision
            eng.select_block_size(pick_block_size(st["n_events"], p_total, pairs=pairs_on))
        elif adaptive_mt:
            eng.select_block_size(pick_block_size_mt(st["n_events"], p_total, allow_1024=mt_1024))
        elif adaptive_mts:
            eng.select_block_size(1024 if st["n_events"] < MT_1024_CHANGE_FRACTION * p_total else 512)
        acc["launches"] += -(-p_loc // (s["bs"] * max(m_now, 1))) + 1
        acc["grouped"] = max(acc.get("grouped", 0), m_now)
        acc["bytes"] += 4.0 * n_loc * p_loc if a.storage == "dense" else 0.25 * n_loc * p_loc
        s["bs"] = eng.block_size
        # 3-5. pi, marker-effect variance, residual variance (Pi.jl:7-42, variance_components.jl:60-112,151-189)
        if method == "BayesR":
            s["pi"] = rng.dirichlet(st["class_counts"] + 1.0)
            s["G"] = np.float32((st["bayesr_ssq"] + df_g * scale_g) / rng.chisquare(st["bayesr_nnz"] + df_g))
            s["vare"] = np.float32((np.float32(st["resid_ss"][0, 0]) + df_e * scale_e) / rng.chisquare(n + df_e))
        elif t > 1:
            from scipy.stats import invwishart
            s["pi"] = rng.dirichlet(st["state_counts"] + 1.0)
            if os.environ.get("JWAS_BENCH_LOG_STATES") and s["it"] % int(os.environ["JWAS_BENCH_LOG_STATES"]) in (0, 1):
                codes = sum((eng.get_state(k)[2] != 0).astype(np.int64) << k for k in range(t))      # the markers' joint states
                if s["it"] % int(os.environ["JWAS_BENCH_LOG_STATES"]) == 0:
                    log(f"sweep {s['it']}: joint-state counts {[int(v) for v in st['state_counts']]} sweep_ms={st['sweep_ms']:.2f}")
                    s["_codes"] = codes
                elif "_codes" in s:          # how many markers changed their joint state from one sweep to the next
                    log(f"sweep {s['it']}: {int((codes != s['_codes']).sum())} markers changed their joint state since the last sweep "
                        f"({int(((codes != s['_codes']) & (s['_codes'] != (1 << t) - 1)).sum())} of them were not in the all-ones state)")
            if mt_pervar:                    # one InverseWishart(df + 1, scale + b_j b_j') draw per marker (variance_components.jl:181-186):
                eng.sample_marker_covariances(df_g + 1.0, scale_g, seed=a.seed, iteration=s["it"], marker_offset=lo)   # on the device, from the resident beta
                s["Gmat"] = None             # (the next sweep uses the resident covariances)
            else:
                S = scale_g + st["beta_ss"]
                s["G"] = np.asarray(invwishart.rvs(df=df_g + p_total, scale=(S + S.T) / 2, random_state=rng), dtype=np.float32).reshape(t, t)
            S = scale_e + st["resid_ss"]
            s["vare"] = np.asarray(invwishart.rvs(df=df_e + n, scale=(S + S.T) / 2, random_state=rng), dtype=np.float32).reshape(t, t)
        else:
            nl = st["sum_delta"][0]
            if estimate_pi:
                s["pi"] = float(rng.beta(p_total - nl + 1.0, nl + 1.0))
            if not refbench:                                         # refbench: estimate_variance = false for the markers
                s["G"] = np.float32((np.float32(st["alpha_ss"][0, 0]) + df_g * scale_g) / rng.chisquare(nl + df_g))
            s["vare"] = np.float32((np.float32(st["resid_ss"][0, 0]) + df_e * scale_e) / rng.chisquare(n + df_e))
        acc["sweep_ms"] += st["sweep_ms"]
        acc["events"] += st["n_events"]
        return st

    def barrier():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    # chain burn-in (setup): only when the warm-up alone would not reach the steady state
    nburn = a.chain if a.chain > 0 else max(0, a.burnin - a.warmup)
    chain_rec = None
    if a.chain > 0:
        # --chain: every sweep from the start on the clock (wall time per iteration incl. the host draws; each step ends with the
        # statistics' device-to-host copy, i.e. synchronised)
        rec = {"wall_ms": [], "sweep_ms": [], "events": [], "in_model": [], "bs": []}
        barrier()
        tc0 = time.perf_counter()
        for _ in range(nburn):
            t1 = time.perf_counter()
            st_ = step()
            rec["wall_ms"].append(1e3 * (time.perf_counter() - t1))
            rec["sweep_ms"].append(float(st_["sweep_ms"]))
            rec["events"].append(float(st_["n_events"]))
            rec["in_model"].append(float(st_["class_counts"][1:].sum()) if method == "BayesR" else
                                   (float(p_total - st_["state_counts"][0]) if t > 1 else float(st_["sum_delta"][0])))
            rec["bs"].append(int(state["bs"]))
        barrier()
        chain_total = time.perf_counter() - tc0
        W = 100
        wm = np.asarray(rec["wall_ms"])
        win = lambda key, f=np.mean: [float(f(np.asarray(rec[key][i:i + W]))) for i in range(0, nburn, W)]      # noqa: E731
        chain_rec = {"sweeps": nburn, "chain_total_s": chain_total, "mean_ms": float(wm.mean()), "worst_sweep_ms": float(wm.max()),
                     "worst_sweep_index": int(wm.argmax()) + 1, "window": W,
                     "window_mean_ms": win("wall_ms"), "window_max_ms": win("wall_ms", np.max), "window_device_sweep_ms": win("sweep_ms"),
                     "window_events_per_sweep": win("events"), "window_markers_in_model": win("in_model"),
                     "window_block_size": [int(np.bincount(np.asarray(rec["bs"][i:i + W])).argmax()) for i in range(0, nburn, W)]}
    else:
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

    per_rank_sweep_ms = [acc["sweep_ms"] / a.steps]
    if world > 1:
        box = [None] * world
        torch.distributed.all_gather_object(box, per_rank_sweep_ms[0])
        per_rank_sweep_ms = [float(v) for v in box]
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
        bs_now = state["bs"]
        achieved = bytes_per_launch / 1e9 / (avg_launch_us * 1e-6)
        variant = (f"{a.mt_method}/{a.mt_prior}" + ("/walk" if (a.no_section_solve and mt_big and t <= 3 and bs == 256) else "")) if t > 1 else None   # (config 4: sampler family, prior, chain form)
        m_used = int(acc.get("grouped", 0))
        traffic, traffic_src = traffic_from_profiles(wl, n, p_total, bs_now, a.storage, world, a.pi_fixed, variant, m_used)
        if t == 1 and method == "BayesC":
            in_model = float(last["sum_delta"][0])
        elif method == "BayesR":
            in_model = float(last["class_counts"][1:].sum())
        else:
            in_model = float(p_total - last["state_counts"][0])
        desc = {
            "config2": f"single-trait BayesC, {n} individuals x {p_total} SNPs, " + ("2-bit packed genotypes (decoded to fp32 on the fly)" if a.storage == "packed2bit" else "fp32 dense genotypes") + (", pi0=0.95 estimated" if a.pi_fixed is None else f", pi={a.pi_fixed} fixed (estimatePi=false)"),
            "config3": f"single-trait BayesR (4-class mixture, gamma 0/.01/.1/1, pi estimated), {n} x {p_total}, fp32 dense genotypes",
            "config4": f"3-trait {'BayesB (one effect covariance per marker, redrawn on the device each iteration)' if a.mt_method == 'BayesB' else 'BayesC'} sampler I, {n} x {p_total}, fp32 dense, R and G inverse-Wishart on host, 8-state pi estimated, start: " + ("all-ones state (reference default)" if a.mt_prior == "default" else "0.95 on the null state"),
            "config5shard": f"single-step shaped BayesC, {n} rows ({n_gen} integer-coded + {n - n_gen} real-valued imputed) x {p_arg} SNPs per GPU ({p_total} in total), fp32 dense",
            "refbench": f"reference benchmark shape (jwas_nonblock_benchmark.jl): BayesC, {n} x {p_total}, X~U[0,1) fp32 uncentred, y~N(0,1), Pi=0 fixed, marker variance fixed",
        }[wl]
        out = {
            "metric": "MCMC iters/sec (full marker sweep)", "value": a.steps / elapsed, "unit": "iterations/s",
            "n_gpus": comm_world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": ms_per_step,
            "higher_is_better": True, "scaling": "weak" if weak else "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": desc, "name": wl, "variant": variant, "storage": a.storage,
                       "n": n, "p": p_total, "block_size": bs_now, "block_policy": "adaptive 512/1024" if adaptive else (("256 while dense / 512 once sparse" + (" / 1024 below 0.5 % turnover" if mt_1024 else "")) if adaptive_mt else ("512 / 1024 below 0.5 % turnover" if adaptive_mts else "fixed")),
                       "blocks_per_launch": (m_used if m_used >= 2 else 1), "group_setup_s": group_setup_s,
                       "parallelism": (f"{'row' if rows_mode else 'marker'}-shard x{world}" + (" (exact chain of the pooled data; one all-reduce of the block RHS per block launch)" if rows_mode else " (one all-reduce of the residual delta per sweep; residual resident in HBM)")) if world > 1 else "single GPU",
                       "ranks_reported_by_communicator": comm_world,
                       "device_sweep_ms": acc["sweep_ms"] / a.steps, "per_rank_device_sweep_ms": per_rank_sweep_ms, "events_per_sweep": acc["events"] / a.steps,
                       "markers_in_model": in_model, "setup_s": setup_s,
                       "host_ms_per_step": ms_per_step - acc["sweep_ms"] / a.steps,
                       "sharded_path": bool(getattr(shard, "_lib_comm", False)),
                       "chain_sweeps_before_timing": nburn + a.warmup},
            "roofline": {"bound": "hbm", "kernel": ("k_group_step" if m_used >= 2 else "k_block_step"), "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
                         "bytes_per_launch": bytes_per_launch, "avg_launch_us": avg_launch_us, "launches_timed": launches},
        }
        if chain_rec is not None:
            out["chain"] = chain_rec
        if world == 1 and not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(a, eng, method, t, n, p_total, min(a.cpu_sample_markers, p_loc), Y, state, refbench)
        nvia = a.via_api if a.via_api >= 0 else (20 if (wl == "config2" and world == 1 and a.pi_fixed is None and a.storage == "dense") else 0)
        if world == 1 and nvia > 0 and bayesc and t == 1 and not refbench:
            out["via_api"] = via_api(eng, Y[0], n, p_total, a, nvia)
    eng.close()
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()
    elif a.one_rank_comm:
        torch.distributed.destroy_process_group()
    if rank == 0:
        print(json.dumps(out), flush=True)
    if _ABANDONED_THREAD:
        os._exit(0)                                  # a CPU-baseline leg is still stuck in native code: do not wait for it


_ABANDONED_THREAD = False


def via_api(eng, y, n, p, a, nsteps):
    """The same workload through the package's user API: device_genotypes() (the matrix already resident on the GPU, like
    the reference's stream_backend handle on Genotypes) -> build_model -> runMCMC; iterations/s over the last `nsteps`
    iterations from the per-iteration timestamps runMCMC records (every iteration ends with a stream synchronisation)."""
    import shutil
    import tempfile
    import pandas as pd
    import jwas_jl_amd as J
    warm = 30
    geno = J.device_genotypes(eng, method="BayesC", Pi=0.95 if a.pi_fixed is None else a.pi_fixed, estimatePi=a.pi_fixed is None)
    model = J.build_model("y = intercept + geno", genotypes={"geno": geno})
    ph = pd.DataFrame({"ID": geno.obsID, "y": y})
    folder = tempfile.mkdtemp(prefix="jwas_bench_")
    try:
        out = J.runMCMC(model, ph, chain_length=warm + nsteps, burnin=warm, seed=a.seed, outputEBV=False,
                        output_samples_frequency=warm + nsteps + 1, output_folder=os.path.join(folder, "results"), printout_model_info=False,
                        blocks_per_launch=eng.blocks_per_launch(1024))      # (the chain-length rule would pick 0 for this short run)
    finally:
        shutil.rmtree(folder, ignore_errors=True)
    ts = out["_timing"]["iteration_end_s"]
    el = ts[-1] - ts[-1 - nsteps]
    return {"value": nsteps / el, "unit": "iterations/s", "steps": nsteps, "warmup": warm, "ms_per_step": 1e3 * el / nsteps,
            "what": "runMCMC(model, df; chain_length, burnin, seed) on the resident matrix: location-parameter Gibbs step, sweep, pi / "
                    "variance draws and running means as the API does them"}


def cpu_baseline(a, eng, method, t, n, p_total, p_sub, Y, state, refbench):
    """The CPU oracle's NON-BLOCK sweep of the same sampler (oracle/jwas_oracle.c: orc_time_sweeps_team -- per marker fp32
    dot, scalar update, conditional fp32 axpy: the reference's operation order, BayesABC.jl:60-80 / BayesR.jl:45-97 /
    MTBayesABC.jl:57-127) on the first p_sub markers of the same matrix, >= 2 sweeps, scaled linearly in p (the reference's
    own projection device, benchmarks/streaming_large_benchmark.jl:161-184).  Timed with 1 thread, a 16-thread team and all
    logical cores (rows of every dot / axpy split over a persistent team, one spin barrier per marker -- the best a threaded
    level-1 BLAS can do); the fastest is reported with its thread count."""
    os.environ.setdefault("OMP_PROC_BIND", "close")
    os.environ.setdefault("OMP_PLACES", "cores")
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle as O
    X = eng.get_columns(0, p_sub)
    xpx = O.xpx(X, O.ACC_F32)
    ncores = os.cpu_count() or 1
    kind = {"BayesC": 0, "BayesR": 1, "MTBayesC": 2}[method]
    vare = np.atleast_2d(np.asarray(state["vare"], dtype=np.float32))
    G = np.atleast_2d(np.asarray(state["G"], dtype=np.float32))
    if method == "BayesR":
        prior = np.asarray(state["pi"], dtype=np.float64)
    elif t > 1:
        with np.errstate(divide="ignore"):
            prior = np.log(np.asarray(state["pi"], dtype=np.float64))
    else:
        prior = np.array([0.0 if refbench else 0.95])
    R0 = np.ascontiguousarray(Y - Y.mean(axis=1, keepdims=True), dtype=np.float32)

    import threading
    res, notes = {}, []
    # CPUs this process may actually use: the container's CFS quota (cgroup cpu.max) caps it below the logical core count
    # on the GPU boxes (1 600 000 / 100 000 = 16 CPUs of 256 logical); threads beyond the quota only burn it and get the
    # whole process throttled for the rest of every 100 ms period
    usable = ncores
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            quota = float(txt[0]) if txt[0] != "max" else -1.0
            period = float(txt[1]) if len(txt) > 1 else float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if quota > 0:
                usable = max(1, min(ncores, int(quota // period)))
                notes.append(f"cgroup CPU quota {quota:.0f}/{period:.0f} us = {usable} usable CPUs")
            break
        except (OSError, ValueError, IndexError):
            continue
    # legs: 1 thread; a persistent team on all USABLE CPUs (one spin barrier per marker); when no quota hides cores, also
    # ALL logical cores with one fork/join per dot / axpy -- what the reference's setting (BLAS threads = cores,
    # jwas_nonblock_benchmark.jl:21-22) does per call
    team = min(usable, 64)
    legs = [(1, 1, "team")] + ([(team, team, "team")] if team > 1 else []) + ([(ncores, -ncores, "fork-join")] if (ncores > 1 and usable == ncores) else [])
    for threads, arg, how in legs:
        al = np.zeros((t, p_sub), dtype=np.float32)
        be = np.zeros((t, p_sub), dtype=np.float32)
        de = np.ones((t, p_sub), dtype=np.int32 if method == "BayesR" else np.float32)
        box = {}

        def leg():
            # up to 20 sweeps, bounded by --cpu-seconds of wall time per leg (a leg that does not finish a sweep in that
            # time is scaled from the marker updates it completed)
            box["r"] = O.time_sweeps_team(kind, X, xpx, R0.copy(), al, be, de, vare, G, prior, a.seed, 20, arg, max_seconds=a.cpu_seconds)
        th = threading.Thread(target=leg, daemon=True)       # (guard: a leg that does not come back is abandoned, not waited for)
        th.start()
        th.join(timeout=4 * a.cpu_seconds + 30)
        if th.is_alive() or "r" not in box:
            notes.append(f"{threads} threads ({how}): did not return within {4 * a.cpu_seconds + 30:.0f} s, abandoned")
            global _ABANDONED_THREAD
            _ABANDONED_THREAD = True
            break
        tt, done = box["r"]
        done = max(int(done), 1)
        res[threads] = (tt * p_sub / done, done / p_sub, how)
        log(f"cpu baseline {threads} thread(s) [{how}]: {res[threads][0]:.3f} s per {p_sub}-marker sweep ({res[threads][1]:.2f} sweeps in {tt:.1f} s)")
    best = min(res, key=lambda k: res[k][0])
    per_sweep_full = res[best][0] * p_total / p_sub
    detail = "; ".join([f"{k} thread(s) [{v[2]}]: {v[0] * p_total / p_sub:.2f} s/sweep ({v[1]:.2f} sweeps of the sample timed)" for k, v in sorted(res.items())] + notes)
    return {"value": 1.0 / per_sweep_full, "unit": "iterations/s", "cores": best, "kind": "port",
            "sample": (f"non-block {method} sweeps over the first {p_sub} of {p_total} markers (n={n}), <= {a.cpu_seconds:.0f} s per leg, scaled linearly in p; "
                       f"{detail}; host has {ncores} logical cores, {usable} usable; sweep only, host updates excluded")}


if __name__ == "__main__":
    main()
