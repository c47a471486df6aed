"""Host side of one MCMC chain: everything of MCMC_BayesianAlphabet (MCMC/MCMC_BayesianAlphabet.jl:4-447)
that is NOT the marker sweep, driving a sweep engine (jwas.jl_amd.engine.HipEngine) for step 2.

Per iteration (the reference's order, MCMC_BayesianAlphabet.jl:184-421):
  1. location parameters: ycorr += X sol ; rhs = X'ycorr ; single-site Gibbs ; ycorr -= X sol   (:196-220, host)
  2. marker effects: engine.sweep(...)                                                      (:224-290, DEVICE)
  3. pi  ~ Beta / Dirichlet from the sweep's counts                                          (:294-317, host)
  4. marker effect variance from alpha'alpha / ssq / beta'beta                               (:321-326, host)
  5. residual variance from r'r                                                              (:363-370, host)
  6. every output_samples_frequency after burn-in: running means (device for markers)       (:399-413)
The host consumes only O(n) + O(p)-reducible quantities from the device.
"""
import os
import time

import numpy as np

BAYESR_GAMMA = np.array([0.0, 0.01, 0.1, 1.0])
DEVICE_BLOCK_SIZES = (64, 128, 256, 512, 1024)


def _supported_block(b):
    return min(DEVICE_BLOCK_SIZES, key=lambda s: abs(s - b))


class _Running:
    """mean += (x-mean)/k ; mean2 += (x^2-mean2)/k   (output.jl:556-560)"""

    def __init__(self, like):
        self.mean = np.zeros_like(np.asarray(like, dtype=np.float64))
        self.mean2 = np.zeros_like(self.mean)

    def add(self, x, k):
        x = np.asarray(x, dtype=np.float64)
        self.mean += (x - self.mean) / k
        self.mean2 += (x * x - self.mean2) / k

    def sd(self):
        return np.sqrt(np.abs(self.mean2 - self.mean ** 2))          # output.jl:112,133


def _design(model, df, ids_col):
    """Fixed-effect incidence matrices per trait (build_MME.jl:183-290, dense): intercept = ones,
    covariate = the column, factor = one 0/1 column per level."""
    cols, labels = [], []
    for tl in model.modelTerms:
        Xk, lab = [], []
        for term in tl:
            if term.kind == "intercept":
                Xk.append(np.ones((len(df), 1)))
                lab.append((term.trait, "intercept", "intercept"))
            elif term.kind == "covariate":
                Xk.append(df[term.name].to_numpy(dtype=np.float64)[:, None])
                lab.append((term.trait, term.name, term.name))
            else:
                if term.name not in df.columns:
                    raise ValueError(f"{term.name} is not found in the phenotype data (genotype terms must be "
                                     "Genotypes objects visible to build_model).")
                lev = df[term.name].astype(str)
                for lv in sorted(lev.unique()):
                    Xk.append((lev == lv).to_numpy(dtype=np.float64)[:, None])
                    lab.append((term.trait, term.name, lv))
        cols.append(np.hstack(Xk) if Xk else np.zeros((len(df), 0)))
        labels.append(lab)
    return cols, labels


def genetic2marker(Mi, pi, method, t=1):
    """Marker-effect (co)variance implied by the genetic variance and pi (tools4genotypes.jl:426-478)."""
    p = Mi.nMarkers
    Vg = np.asarray(Mi.genetic_variance.val, dtype=np.float64)
    if t > 1:                                                          # :426-438 (Dict Pi)
        pi_arr = np.asarray(pi, dtype=np.float64)
        denom = np.zeros((t, t))
        for i in range(t):
            for j in range(t):
                sel = [s for s in range(1 << t) if (s >> i) & 1 and (s >> j) & 1]
                denom[i, j] = Mi.sum2pq * pi_arr[sel].sum()
        return Vg / denom
    if method == "BayesR":                                             # :461-467
        pv = np.asarray(pi, dtype=np.float64)
        if pv.shape != (4,):
            raise ValueError("BayesR Pi must have length 4.")
        denom = Mi.sum2pq * float((BAYESR_GAMMA * pv).sum())
        if not denom > 0:
            raise ValueError("BayesR implied variance denominator must be positive.")
        return float(Vg) / denom
    if np.ndim(pi) == 1:                                               # :468-475
        if len(pi) != p:
            raise ValueError(f"BayesC marker-level Pi must have length {p}.")
        af = np.asarray(Mi.alleleFreq, dtype=np.float64)
        denom = float((2 * af * (1 - af) * (1 - np.clip(np.asarray(pi, dtype=np.float64), 0, 1))).sum())
        if not denom > 0:
            raise ValueError("BayesC implied variance denominator must be positive.")
        return float(Vg) / denom
    return float(Vg) / ((1 - float(pi)) * Mi.sum2pq)                   # :457-459


ADAPTIVE_CHANGE_FRACTION = float(os.environ.get("JWAS_ADAPTIVE_FRACTION", "0.0125"))      # measured crossover of block 512 vs 1024


ADAPTIVE_CHANGE_FRACTION_PAIRS = float(os.environ.get("JWAS_ADAPTIVE_FRACTION_PAIRS", "0.009"))    # ... when the 512-marker sweeps run as ping-pong pairs


def pick_block_size(n_events, p, small=512, large=1024, pairs=False):
    """Block size of the next sweep from the number of markers whose effect changed in the last one.  pairs: the small size runs as
    ping-pong pairs (pingpong_pairs_for_chain), which moves the crossover down (config 3's chain: 24 ms per sweep on 512-marker pairs at
    8 700 changes per sweep, 26.5 ms on 4 x 1024-marker launches at 7 500, 22.6 ms at 4 700)."""
    return large if n_events < (ADAPTIVE_CHANGE_FRACTION_PAIRS if pairs else ADAPTIVE_CHANGE_FRACTION) * p else small


# Grouped launches (jwas_hip_setup_groups / jwas_sweep_params.group_launch): blocks per launch of the step kernel on the LARGE block
# size of a single-trait chain -- the size the adaptive policy selects once few markers change per sweep, where a launch's fixed
# cost (~3.5 us of a 32 us launch at 50 000 x 1024) is what separates the sweep from the copy rate.  0 = off.
GROUPED_BLOCKS_PER_LAUNCH = 4


def grouped_blocks_for_chain(chain_length):
    """Blocks per grouped launch a chain of this length pays for: the group cross-Grams are set-up work (at 50 000 x 600 000:
    +1.7 s for 2 blocks per launch, +5 s for 4) against 0.6 / 0.9 ms saved per sweep of the sparse steady state."""
    # (measured break-even, round 6: 4 blocks per launch + 4.1 s of set-up / 1.07 ms per sweep = 3 800 sweeps; 2 blocks: + 1.4 s / 0.6 ms
    # = 2 300 -- the thresholds leave a margin of 1.3x, they were 8 000 / 3 000)
    return GROUPED_BLOCKS_PER_LAUNCH if chain_length >= 5000 else (2 if chain_length >= 3000 else 0)


def pingpong_pairs_for_chain(method, estimate_pi, chain_length):
    """Pairs of 512-marker blocks per launch (ping-pong samplers: csrc/sweep.hpp k_group_step, SamplerArgs::pp_role) for the
    HIGH-TURNOVER sweeps of a single-trait chain -- the ones the adaptive policy runs on 512-marker blocks because the sampler is the
    critical path: BayesR sheds its markers over hundreds of sweeps, a fixed pi keeps ~(1 - pi) p markers in the model for ever
    (config 3: 29.5 -> 26.8 ms per sweep, fixed pi = 0.95: 29.4 -> 27.0).  BayesC with pi estimated leaves that regime after ~25
    sweeps: not worth the pair cross-Grams (4 p 1024 bytes, 0.7 s at 50 000 x 600 000)."""
    # (4 blocks per launch, one sampler workgroup per block: config 3 25.2 ms, fixed pi 26.1 ms per sweep against 26.6 / 27.0 with pairs,
    # for + 2.5 GB and + 1.1 s of cross-Grams: pays from ~800 high-turnover sweeps on -- a fixed pi stays in the regime for ever, config 3's
    # BayesR chain runs its first ~1 300 sweeps on 512-marker blocks)
    if chain_length >= 2000 and (method == "BayesR" or not estimate_pi):
        return 4
    return 2 if (chain_length >= 300 and (method == "BayesR" or not estimate_pi)) else 0


def grouped_launch_size(method, ntraits, row_shards, block_size, groups=GROUPED_BLOCKS_PER_LAUNCH, dense_prior=False):
    """Block size on which grouped launches are set up (0: none): single-trait BayesA/B/C/R chains with a sparse prior on uniform
    blocks of one GPU's markers (marker shards included, row shards not), groups * block_size <= 4096."""
    if groups not in (2, 4) or ntraits != 1 or row_shards or dense_prior or method not in ("BayesC", "BayesB", "BayesA", "BayesR"):
        return 0
    if block_size < 512 or groups * block_size > 4096:
        return 0
    return int(block_size)


MT_SPARSE_CHANGE_FRACTION = float(os.environ.get("JWAS_MT_SPARSE_FRACTION", "0.1"))          # the share of markers changing per sweep below which a dense-start multi-trait chain is sparse


MT_1024_CHANGE_FRACTION = float(os.environ.get("JWAS_MT_1024_FRACTION", "0.005"))          # ... below which it runs 1024-marker blocks (skip and verify)


def mt_1024_allowed(t, p, per_marker_cov=False):
    """1024-marker multi-trait blocks: the block's draws and constants must fit LDS (block x traits <= 3072: up to three traits,
    no covariance per marker) and there must be a few of them."""
    return bool(1024 * t <= 3072 and not per_marker_cov and p > 4 * 1024)


def pick_block_size_mt(n_events, p, allow_1024=False):
    """Multi-trait sampler I that STARTS dense (the reference's default prior, every marker in the model): 256-marker blocks
    (dense_big_mt / Rule T) while most markers change every sweep; with Pi estimated such a chain moves on to a sparse steady
    state, where the speculative rounds want 512-marker blocks (3.98 vs 5.12 ms per sweep at 20k x 100k x 3)
    and, once a block holds only a handful of candidates (< 0.5 % of the markers change per sweep: most 64-marker sub-blocks are left
    to the helper waves, sampler_role_mt's skip and verify), 1024-marker blocks: half the launches and fronts (3.35 vs 3.83 ms)."""
    if allow_1024 and n_events < MT_1024_CHANGE_FRACTION * p:
        return 1024
    return 512 if n_events < MT_SPARSE_CHANGE_FRACTION * p else 256


def _impute_missing_residuals(res, observed, R0, rng):
    """sampleMissingResiduals (residual.jl:52-73), in place on the per-trait residual vectors `res`: for every missing
    pattern the missing residuals are drawn from their conditional distribution given the observed ones,
    e_m | e_o ~ N(Rc Ro^-1 e_o, Rmm - Rc Ro^-1 Rc').  Returns the per-record inverse residual covariance of mkRi/getRi
    (residual.jl:2-44): inv(R0[o, o]) embedded in a t x t matrix of zeros."""
    n, t = observed.shape
    Ri_rows = np.empty((n, t, t))
    codes = observed @ (1 << np.arange(t))
    full = (1 << t) - 1
    for code in np.unique(codes):
        rows = np.nonzero(codes == code)[0]
        o = np.array([(code >> k) & 1 for k in range(t)], dtype=bool)
        Ro_inv = np.linalg.inv(R0[np.ix_(o, o)])
        RZ = np.zeros((t, t))
        RZ[np.ix_(o, o)] = Ro_inv
        Ri_rows[rows] = RZ
        if code == full:
            continue
        m = ~o
        Rc = R0[np.ix_(m, o)]
        U = np.linalg.cholesky(R0[np.ix_(m, m)] - Rc @ Ro_inv @ Rc.T).T          # upper factor, as cholesky(...).U
        eo = np.stack([res[k][rows] for k in np.nonzero(o)[0]], axis=1)          # rows x n_obs
        em = eo @ Ro_inv @ Rc.T + rng.standard_normal((len(rows), int(m.sum()))) @ U
        for c, k in enumerate(np.nonzero(m)[0]):
            res[k][rows] = em[:, c]
    return Ri_rows


def _inverse_wishart_batch(rng, df, scale):
    """One InverseWishart(df, scale_j) draw per matrix of the batch (m x t x t), Bartlett's decomposition: with
    scale_j = C C' (Cholesky) and A lower-triangular (A_ii = sqrt(chi2(df - i)), A_ik ~ N(0,1) below the diagonal),
    W = C'^-1 A A' C^-1 ~ Wishart(df, scale_j^-1) and G = W^-1 = K K' with K' = A^-1 C' (one batched Cholesky and a forward
    substitution vectorised over the batch; the reference draws each marker's matrix with rand(InverseWishart(df, scale)),
    variance_components.jl:181-186)."""
    m, t, _ = scale.shape
    C = np.linalg.cholesky(scale)
    A = np.zeros((m, t, t))
    for i in range(t):
        A[:, i, i] = np.sqrt(rng.chisquare(df - i, size=m))
        for k in range(i):
            A[:, i, k] = rng.standard_normal(m)
    Kt = np.empty((m, t, t))
    Ct = C.transpose(0, 2, 1)
    for i in range(t):
        acc = Ct[:, i, :].copy()
        for k in range(i):
            acc -= A[:, i, k, None] * Kt[:, k, :]
        Kt[:, i, :] = acc / A[:, i, i, None]
    G = Kt.transpose(0, 2, 1) @ Kt
    return (G + G.transpose(0, 2, 1)) / 2


def _gibbs(A, x, b, rng, vare=None):
    """One sweep of the single-site Gibbs sampler on the MME (iterative_solver/solver.jl:143-162)."""
    for i in range(len(x)):
        if A[i, i] != 0.0:
            invlhs = 1.0 / A[i, i]
            mu = invlhs * (b[i] - A[:, i] @ x) + x[i]
            x[i] = rng.standard_normal() * np.sqrt(invlhs * (vare if vare is not None else 1.0)) + mu


def run_chain(model, df, *, chain_length, burnin, output_samples_frequency, seed, starting_value,
              fast_blocks, independent_blocks=False, heterogeneous_residuals=False, outputEBV, output_heritability=True, output_folder, printout_frequency, memory_guard, memory_guard_ratio,
              missing_phenotypes, device, block_size, gram_mode, engine, printout_model_info,
              output_samples_for_all_parameters, double_precision=False, blocks_per_launch=None):
    import pandas as pd
    Mi = model.M[0]
    t = model.nModels
    method = Mi.method
    # runMCMC(double_precision=true) (JWAS.jl:349-366): genotypes, G, alpha -> Float64; everything the chain holds follows.
    # ftype is the element type of the run (the reference's Float32 default, or Float64).
    ftype = np.float64 if double_precision else np.float32
    if not double_precision and getattr(Mi, "genotypes", None) is not None and getattr(Mi.genotypes, "dtype", None) == np.float64:
        raise NotImplementedError("Float64 genotypes (get_genotypes(double_precision=true)) run with runMCMC(double_precision=true); "
                                  "the mixed Float64 / Float32 chain stays on the reference")
    if t > 1 and method not in ("BayesC", "RR-BLUP", "BayesB", "BayesA"):
        raise NotImplementedError("multi-trait device path implements BayesC (Gibbs samplers I, II and the constraint=true "
                                  "megaBayesABC! path) and BayesA/B (the same three); other methods stay on the reference")
    mega = t > 1 and bool(Mi.G.constraint)                              # megaBayesABC! (MCMC_BayesianAlphabet.jl:233-234)
    # Methods the reference never runs block-wise: megaBayesABC! / megaBayesC0! (constraint = true), BayesC0! / MTBayesC0!
    # (RR-BLUP) and BayesL! are ONE plain pass over the markers per outer iteration whatever fast_blocks says -- although
    # chain_length has already been divided by the block size (MCMC_BayesianAlphabet.jl:233-262, JWAS.jl:308-316).
    plain_pass = mega or method in ("RR-BLUP", "BayesL")
    if t == 1 and (Mi.G.constraint or model.R.constraint):
        raise ValueError("constraint==true is for multi-trait only")     # input_data_validation.jl:534-535,550-551
    if not isinstance(starting_value, bool) or starting_value:
        raise NotImplementedError("starting values for location parameters stay on the reference; marker starting "
                                  "values go through get_genotypes(starting_value=...)")
    seed_int = 0 if seed is False else int(seed)
    rng = np.random.default_rng(seed_int)                                # host-side draws (JWAS.jl:239-241)

    # ---- align phenotypes and genotypes (input_data_validation.jl:198-294, tools4genotypes.jl:288-323)
    idcol = df.columns[0]
    ph = df.copy()
    ph[idcol] = ph[idcol].astype(str)
    for tr in model.lhsVec:
        if tr not in ph.columns:
            raise ValueError(f"Phenotypes for {tr} are not found in the data.")
    complete = np.ones(len(ph), dtype=bool)
    for tr in model.lhsVec:
        complete &= np.isfinite(ph[tr].to_numpy(dtype=np.float64))
    if t > 1:
        # individuals whose phenotypes are missing for ALL traits are removed (input_data_validation.jl:396-404);
        # partially missing records need the reference's residual imputation (residual.jl:15-73)
        anyobs = np.zeros(len(ph), dtype=bool)
        for tr in model.lhsVec:
            anyobs |= np.isfinite(ph[tr].to_numpy(dtype=np.float64))
        if (anyobs & ~complete).any() and not missing_phenotypes:
            raise ValueError("phenotypes are missing for some traits of some individuals; missing_phenotypes=false does not allow that")
        usable = anyobs                          # partially observed records stay; their residuals are imputed every iteration
    else:
        usable = complete
    stream = getattr(Mi, "storage_mode", "dense") == "stream"
    devres = getattr(Mi, "storage_mode", "dense") == "device"          # matrix already resident on the GPU (api.device_genotypes)
    if devres:
        if not complete.all() or list(ph[idcol]) != list(Mi.obsID):
            raise ValueError("device-resident genotypes require exact genotype/phenotype ID match and order "
                             "(as storage=:stream does, JWAS.jl:388-398).")
        if engine is not None and engine is not Mi.device_backend:
            raise ValueError("the genotypes are resident on a different engine")
        engine = Mi.device_backend
        X = None
        n, p = Mi.nObs, Mi.nMarkers
    elif stream:
        # the packed payload goes from the file straight to HBM and is never re-ordered: like the reference's stream
        # mode, phenotype IDs must match the genotype IDs exactly and in order (JWAS.jl:388-398)
        if not complete.all() or list(ph[idcol]) != list(Mi.obsID):
            raise ValueError("storage=:stream MVP requires exact genotype/phenotype ID match and order. "
                             "Please reorder phenotypes to match genotype IDs.")
        print("storage=:stream is enabled; genotype alignment is skipped and original ID order is used.")
        X = None
        n, p = Mi.nObs, Mi.nMarkers
    else:
        geno_index = {g: i for i, g in enumerate(Mi.obsID)}
        keep = usable & ph[idcol].isin(geno_index).to_numpy()
        ph = ph.loc[keep].reset_index(drop=True)
        if len(ph) == 0:
            raise ValueError("no individual has both phenotypes and genotypes")
        rows = np.array([geno_index[i] for i in ph[idcol]], dtype=np.int64)
        X = Mi.genotypes if (len(rows) == Mi.nObs and np.array_equal(rows, np.arange(Mi.nObs))) else np.asfortranarray(Mi.genotypes[rows, :])
        n, p = X.shape
    # ---- individuals EBVs are reported for (check_outputID, input_data_validation.jl:143-196): all genotyped
    # individuals unless outputEBV(model, IDs) named a list; IDs without genotypes are dropped with the reference's note
    out_ids, out_rows, out_same = None, None, True
    if outputEBV:
        want = list(Mi.obsID) if getattr(model, "output_ID", False) is False else list(model.output_ID)
        known = set(Mi.obsID)
        if not all(i in known for i in want):
            print("Testing individuals are not a subset of genotyped individuals (complete genomic data,non-single-step). "
                  "Only output EBV for tesing individuals with genotypes.")
            want = [i for i in want if i in known]
        out_ids = want
        out_same = out_ids == list(ph[idcol])             # exactly the training rows, same order: X itself
        if not out_same:
            if stream or devres:
                raise NotImplementedError("storage=:stream reports EBVs for the genotyped individuals in file order "
                                          "(outputEBV(model, IDs) lists stay on the reference)")
            gi = {g: i for i, g in enumerate(Mi.obsID)}
            out_rows = np.array([gi[i] for i in out_ids], dtype=np.int64)
    if not stream and not devres:
        # rows of Mi.genotypes behind Mi.output_genotypes (tools4genotypes.jl:290-296); GWAS() reads them (GWAS.jl:148)
        Mi.output_rows = out_rows if (outputEBV and not out_same) else rows
    with open(os.path.join(output_folder, "IDs_for_individuals_with_phenotypes.txt"), "w") as fh:
        fh.write("\n".join(ph[idcol]) + "\n")
    with open(os.path.join(output_folder, "IDs_for_individuals_with_genotypes.txt"), "w") as fh:
        fh.write("\n".join(Mi.obsID) + "\n")
    Y = np.stack([ph[tr].to_numpy(dtype=ftype) for tr in model.lhsVec])       # t x n
    observed = np.isfinite(Y).T                                                     # n x t: mme.missingPattern (residual.jl:17-21)
    has_missing = not observed.all()
    phenovar = np.array([np.var(Y[k][observed[:, k]].astype(np.float64), ddof=1) for k in range(t)])
    Y = np.where(np.isfinite(Y), Y, ftype(0)).astype(ftype)               # imputed before first use (residual.jl:52-73)
    invw = None
    if heterogeneous_residuals:                                           # build_MME.jl:305-310
        if "weights" not in ph.columns:
            raise ValueError("heterogeneous_residuals=true requires a column named weights in the phenotype data.")
        invw = (1.0 / ph["weights"].to_numpy(dtype=np.float64)).astype(ftype)
        if not np.all(np.isfinite(invw) & (invw > 0)):
            raise ValueError("weights must be positive and finite.")
    w64 = np.ones(len(ph)) if invw is None else invw.astype(np.float64)

    # ---- default priors (input_data_validation.jl:296-350, tools4genotypes.jl:353-478, build_MME.jl:128-141)
    varg = np.diag(phenovar) * 0.5
    vare0 = np.diag(phenovar) * 0.5
    R = model.R
    if R.val is False:
        R.val = ftype(vare0[0, 0]) if t == 1 else vare0.astype(ftype)
        R.scale = float(R.val) * (float(R.df) - 2) / float(R.df) if t == 1 else np.asarray(R.val, dtype=np.float64) * (float(R.df) - t - 1)
    if Mi.G.val is False and Mi.genetic_variance.val is False:
        Mi.genetic_variance.val = varg[0, 0] if t == 1 else varg
    pi = Mi.pi
    if isinstance(pi, dict):
        # the reference's multi-trait Pi: Dict(state vector => probability), e.g. Dict([1.0,0.0] => 0.1, ...)
        # (tools4genotypes.jl:357-373, test/runtests.jl); device order: state index = sum_k delta_k << k
        if t == 1:
            raise ValueError("a Dict Pi is for multi-trait analyses only")
        tab = np.zeros(1 << t)
        for key, val in pi.items():
            key = tuple(float(v) for v in key)
            if len(key) != t or any(v not in (0.0, 1.0) for v in key):
                raise ValueError(f"Pi keys must be 0/1 vectors of length {t} (got {key})")
            tab[sum(1 << k for k in range(t) if key[k] == 1.0)] = float(val)
        if abs(tab.sum() - 1.0) > 1e-6:
            raise ValueError("Summation of probabilities of Pi is not equal to one.")       # input_data_validation.jl
        pi = tab
    if (t > 1 and getattr(Mi, "annotations", False) is not False and np.ndim(pi) == 1 and len(pi) == Mi.nMarkers
            and len(pi) != (1 << t) and np.all(np.asarray(pi) == np.asarray(pi)[0])):
        # get_genotypes expanded a scalar Pi to one value per marker for annotated BayesC before the number of traits was
        # known (readgenotypes.jl:111-150); the multi-trait set-up reads the scalar (annotation_setup.jl:106-118)
        pi = float(np.asarray(pi)[0])
    if t > 1 and (np.isscalar(pi) and pi == 0.0):                      # tools4genotypes.jl:357-373
        pi = np.zeros(1 << t)
        pi[(1 << t) - 1] = 1.0
        Mi._pi_was_default = True
    if method == "BayesR" and np.isscalar(pi) and pi == 0.0:           # :375-377
        pi = np.array([0.95, 0.03, 0.015, 0.005])
    if method == "BayesA":                                             # input_data_validation.jl:33-36
        method, Mi.estimatePi = "BayesB", False
        pi = 0.0 if t == 1 else np.eye(1, 1 << t, (1 << t) - 1).ravel()  # every marker in the model for every trait
    lasso = method == "BayesL"
    if lasso:
        # Bayesian LASSO (BayesL!, BayesC0L.jl:25-47): every marker in the model, effect variance G*gamma_j with the
        # gamma_j updated by Metropolis-Hastings on the host (sampleGammaArray!, variance_components.jl:191-203).  Its full
        # conditional lhs = x'x + (vare/G)/gamma_j is the device's BayesB update with pi = 0 and var_j = G*gamma_j.
        if t > 1:
            raise NotImplementedError("multi-trait BayesL stays on the reference")
        if not (np.isscalar(pi) and (pi is False or pi == 0.0)):
            print("BayesL runs with π = false.")                        # input_data_validation.jl:24-31
        elif Mi.estimatePi:
            print("BayesL runs with estimatePi = false.")
        method, pi, Mi.estimatePi = "BayesB", 0.0, False
    if method == "RR-BLUP":                                            # input_data_validation.jl:24-31
        if not (np.isscalar(pi) and (pi is False or pi == 0.0)):
            print("RR-BLUP runs with π = false.")
        elif Mi.estimatePi:
            print("RR-BLUP runs with estimatePi = false.")
        # the device's BayesC update with pi = 0 (every marker included) is RR-BLUP's full conditional (api.SUPPORTED_METHODS)
        method, Mi.estimatePi = "BayesC", False
        pi = 0.0 if t == 1 else np.eye(1, 1 << t, (1 << t) - 1).ravel()
    if t == 2 and getattr(Mi, "annotations", False) is not False and not mega:
        # start-row validation of the annotated 2-trait model comes first (finalize_marker_annotation_setup!,
        # annotation_setup.jl:101-121, runs at build_model time in the reference)
        from . import annotations as A_
        A_.bayesc_mt_start_row(0.0 if getattr(Mi, "_pi_was_default", False) else pi)
    if Mi.G.val is False:
        Mi.G.val = genetic2marker(Mi, pi, method, t)
        if (t == 1 and not Mi.G.val > 0) or (t > 1 and (not np.all(np.isfinite(Mi.G.val)) or np.any(np.linalg.eigvalsh(Mi.G.val) <= 0))):
            raise ValueError("Marker effects variance is negative!" if t == 1 else
                             "Marker effects covariance matrix is not postive definite! Please modify the argument: Pi.")
    ann = getattr(Mi, "annotations", False)
    if ann is not False:
        from . import annotations as A_
        if t > 1:                                                      # annotation_setup.jl:101-133
            if t != 2:
                raise ValueError("Annotated multi-trait BayesC currently supports exactly 2 traits.")
            if mega:
                raise NotImplementedError("annotated multi-trait BayesC with constraint=true stays on the reference")
            start_row = A_.bayesc_mt_start_row(0.0 if getattr(Mi, "_pi_was_default", False) else pi)
            ann = Mi.annotations = A_.initialize_bayesc_mt(ann.design_matrix, start_row)
            pi = start_row.copy()
        if stream:
            raise NotImplementedError("marker annotations with storage=:stream stay on the reference")
        if method == "BayesC":                                         # annotation_setup.jl:78-99
            pi = np.array(pi, dtype=np.float64) if np.ndim(pi) == 1 else np.full(p, float(pi))
            A_.initialize_bayesc_single_trait(ann, pi)
        Mi.estimatePi = True
    Gdf = float(Mi.G.df)
    Mi.G.scale = (np.float64(Mi.G.val) * (Gdf - 2) / Gdf) if t == 1 else np.asarray(Mi.G.val, dtype=np.float64) * (Gdf - t - 1)   # :414-418
    Rdf = float(R.df)
    if t > 1 and R.constraint:                                         # R_constraint! (input_data_validation.jl:530-540)
        Rdf -= t
        R.scale = np.diag(np.diag(np.asarray(R.scale, dtype=np.float64)) / (Rdf - 1)) * (Rdf - 2) / Rdf
        R.val = np.diag(np.diag(np.asarray(R.val, dtype=ftype)))
    pi_t = None
    if mega:                                                           # G_constraint! (input_data_validation.jl:543-559)
        Gdf -= t
        Mi.G.scale = np.diag(np.diag(np.asarray(Mi.G.scale, dtype=np.float64)) / (Gdf - 1)) * (Gdf - 2) / Gdf
        Mi.G.val = np.diag(np.diag(np.asarray(Mi.G.val, dtype=ftype)))
        # megaBayesABC! reads one pi per trait (genotypes.pi[i], BayesABC.jl:5).  The reference only holds the joint
        # 2^t table before the first samplePi; the per-trait value used here is its marginal Pr(delta_k = 0).
        pa = np.asarray(pi, dtype=np.float64)
        pi_t = np.array([pa[[s_ for s_ in range(1 << t) if not (s_ >> k) & 1]].sum() for k in range(t)]) if pa.size == (1 << t) else pa
        if pi_t.shape != (t,):
            raise ValueError(f"Pi must hold one value per trait or one per joint state (got {pa.size} entries for {t} traits)")
    sampler = getattr(Mi, "multi_trait_sampler", "I")                   # mt_bayesc_sampler_mode (MTBayesABC.jl:20-25)
    if t > 1 and sampler == "auto":
        # support-based dispatch: a Pi Dict that does not list every joint state (a restricted support, where sampler I
        # cannot move between the listed states) selects the joint-state sampler II
        n_states = len(Mi.pi) if isinstance(Mi.pi, dict) else (1 << t)
        sampler = "I" if n_states == (1 << t) else "II"
    mt_method = "MegaBayesC" if mega else ("MTBayesC_II" if sampler == "II" else "MTBayesC")
    mt_pervar = t > 1 and method == "BayesB"        # multi-trait BayesA/B: one t x t effect covariance per marker
    if mt_pervar:
        if getattr(Mi, "annotations", False) is not False:
            raise NotImplementedError("annotated multi-trait BayesB stays on the reference")
        # constraint = true: megaBayesABC! reads every marker's own diagonal (BayesABC.jl:5)
        mt_method = "MegaBayesB" if mega else ("MTBayesB_II" if sampler == "II" else "MTBayesB")
        if independent_blocks:
            raise NotImplementedError("independent_blocks with multi-trait BayesA/B (one effect covariance per marker) stays on the reference")

    # ---- fast_blocks parsing (JWAS.jl:293-316); device blocks are powers of two >= 64
    nreps = 1
    explicit_partition = None                                          # 0-based starts of a non-uniform explicit partition
    if fast_blocks is not False:
        explicit_starts = False
        if fast_blocks is True:
            want = int(np.floor(np.sqrt(n)))
        elif np.isscalar(fast_blocks):
            want = int(np.floor(fast_blocks))
        else:
            try:
                starts = [int(v) for v in fast_blocks]
            except (TypeError, ValueError):
                raise ValueError("fast_blocks must be false, true, a positive number, or a vector of block start positions.")
            if any(int(v) != v for v in fast_blocks):
                raise ValueError("fast_blocks must be false, true, a positive number, or a vector of block start positions.")
            if len(starts) == 0:                                         # validate_fast_block_starts (JWAS.jl:73-79)
                raise ValueError("fast_blocks block start vector cannot be empty.")
            if starts[0] != 1:
                raise ValueError("fast_blocks block starts must begin with 1.")
            if not all(1 <= v <= p for v in starts):
                raise ValueError("fast_blocks block starts must be within 1:nMarkers.")
            if not all(b_ > a_ for a_, b_ in zip(starts, starts[1:])):
                raise ValueError("fast_blocks block starts must be sorted and unique.")
            sizes = [b_ - a_ for a_, b_ in zip(starts, starts[1:])] + [p - starts[-1] + 1]
            if len(starts) < 2:
                raise ValueError("fast_blocks block size must create at least two block starts.")
            # the reference does not rescale chain_length for explicit starts (JWAS.jl:298-304: block_size = false)
            explicit_starts = True
            if max(sizes) > max(DEVICE_BLOCK_SIZES):
                # a block of more than 1024 markers does not fit the sampler's LDS plan: the nearest legal partition is the
                # given one with every oversized block cut into k = ceil(size / 1024) BALANCED pieces (sizes differ by at
                # most one marker: no sliver of a tail piece), each piece repeated its own size (BayesABC.jl:153) -- a valid
                # chain with MORE within-block sweeps per outer iteration than the request (a block of 1150 markers becomes
                # 575 + 575 repetitions of half-blocks instead of 1150 of the whole); chain_length, the number of saved
                # samples and of hyper-parameter updates are unchanged.  Said out loud, as the reference prints its block
                # size (JWAS.jl:308-316).
                lim = max(DEVICE_BLOCK_SIZES)
                cut = []
                for a_, sz in zip(starts, sizes):
                    k_ = -(-sz // lim)
                    base, extra = divmod(sz, k_)
                    pos = a_
                    for i_ in range(k_):
                        cut.append(pos)
                        pos += base + (1 if i_ < extra else 0)
                print(f"NOTICE: fast_blocks blocks of up to {max(sizes)} markers exceed the device limit of {lim}; "
                      f"running {len(cut)} blocks (oversized blocks cut into balanced pieces of at most {lim} markers, each "
                      f"repeated its own size: the schedule differs from the request's, the number of saved samples does not)")
                starts = cut
                sizes = [b_ - a_ for a_, b_ in zip(starts, starts[1:])] + [p - starts[-1] + 1]
            if len(set(sizes[:-1])) != 1 or sizes[-1] > sizes[0]:
                # NON-UNIFORM partition: the device runs exactly these blocks, each with its own size as repetition count
                # (BayesABC.jl:153); jwas_hip_setup_blocks_explicit
                explicit_partition = np.asarray(starts, dtype=np.int64) - 1
            want = max(sizes) if explicit_partition is not None else sizes[0]
        if want < 1:
            raise ValueError("fast_blocks block size must be at least 1.")
        if want >= p:                                                   # range(1, step=want, stop=p) has one start
            raise ValueError("fast_blocks block size must create at least two block starts.")
        # The schedule AND the partition are the reference's (JWAS.jl:308-312): block starts collect(range(1, step=want,
        # stop=p)), every block repeated its own size (BayesABC.jl:153: the last, shorter block fewer times), and
        # chain_length / want outer iterations -- the same number of hyper-parameter updates and saved samples.
        #   * `want` is one of the device's uniform block sizes (64 ... 1024): the uniform layout IS that partition
        #     (nreps <= 0 = every block its own size);
        #   * any other size <= 1024: the ragged-partition form (jwas_hip_setup_blocks_explicit) with those starts;
        #   * more than 1024 markers per block do not fit the sampler's LDS plan: the nearest legal schedule with a printed
        #     NOTICE (below: blocks of 1024 markers, chain_length / 1024 outer iterations -- fewer saved samples and
        #     hyper-parameter updates per requested iteration than chain_length / want would give)
        #     (independent_blocks alone may ask for more: uniform device blocks of 1024 markers with `want` repetitions
        #     each -- independent blocks are the reference's own approximation).
        if want > max(DEVICE_BLOCK_SIZES) and not explicit_starts and not independent_blocks:
            # fast_blocks = true on more than 1024^2 records, or a number above 1024: the sampler's LDS plan holds the draws,
            # constants and staged Gram rows of at most 1024 markers.  The nearest legal schedule is the reference's own at
            # the largest device size -- blocks of 1024 markers, each repeated 1024 times, chain_length / 1024 outer
            # iterations -- and the run says so where the reference prints its block size (JWAS.jl:308-316).
            print(f"NOTICE: fast_blocks block size {want} exceeds the device limit of {max(DEVICE_BLOCK_SIZES)} markers per block; "
                  f"running BLOCK SIZE {max(DEVICE_BLOCK_SIZES)} (chain_length / {max(DEVICE_BLOCK_SIZES)} outer iterations)")
            want = max(DEVICE_BLOCK_SIZES)
        if not explicit_starts:
            chain_length = int(np.floor(chain_length / want))
        if explicit_partition is None and want not in DEVICE_BLOCK_SIZES and want <= max(DEVICE_BLOCK_SIZES):
            explicit_partition = np.arange(0, p, want, dtype=np.int64)          # = collect(range(1, step=want, stop=p)) - 1
            sizes = [want] * (len(explicit_partition) - 1) + [p - int(explicit_partition[-1])]
        block_size = _supported_block(want)
        nreps = want if (explicit_partition is None and want not in DEVICE_BLOCK_SIZES) else 0   # 0: every block its own size
        if plain_pass:
            nreps = 1            # the reference's schedule for these methods: chain_length / block size plain passes
        if explicit_partition is not None:
            while block_size < want:
                block_size *= 2
            print(f"BLOCK STARTS: {len(explicit_partition)} blocks of {min(sizes)}..{max(sizes)} markers")
        else:
            print(f"BLOCK SIZE: {want}" + (f" (device blocks of {min(block_size, p)} markers)" if min(block_size, p) != want else ""))
    if mt_pervar and block_size is not None and block_size * t > 2048:
        # the markers' own covariances are parked in LDS beside their draws (sampler_mt.hpp): known before anything is loaded
        raise NotImplementedError(f"multi-trait BayesA/B needs fast_blocks * traits <= 2048 on the device (got {block_size} x {t})")
    adaptive = False
    adaptive_mt = False
    adaptive_mts = False
    mt_1024 = False
    section_solve = False
    if block_size is None:
        # Device block size.  Sparse priors (few markers change per sweep): big blocks amortise the per-launch cost.
        # Dense priors (every marker is in the model: Pi = 0 / BayesA / the multi-trait default of all-ones) change
        # every marker every sweep, so the whole block Gram must sit in LDS: 128 x 128 floats.
        if mega:
            dense = float(np.mean(pi_t)) < 0.5
        elif t > 1:
            dense = float(np.asarray(pi, dtype=np.float64)[(1 << t) - 1]) > 0.5
        elif method == "BayesR":
            dense = float(np.asarray(pi, dtype=np.float64)[0]) < 0.5
        else:
            dense = float(np.mean(pi)) < 0.5
        # ... except when the prior includes every marker whatever its rhs (single-trait BayesA / B / C with Pi = 0, RR-BLUP,
        # BayesL): the sampler's dense_big_st path walks 512-marker blocks section by section (diagonal Gram tiles in LDS,
        # the rest applied in parallel) -- four times fewer launches of the streaming role.
        all_in = (t == 1 and method in ("BayesC", "BayesB") and np.ndim(pi) == 0 and float(pi) == 0.0 and not Mi.estimatePi
                  and getattr(Mi, "annotations", False) is False)
        # ... and multi-trait BayesC sampler I with one shared covariance and the shared prior table (the reference's default
        # all-ones prior): 256-marker blocks through dense_big_mt (round 4)
        mt_big = (t > 1 and not mega and mt_method in ("MTBayesC", "MTBayesB") and getattr(Mi, "annotations", False) is False)
        block_size = (512 if all_in else (256 if mt_big else 128)) if dense else 512
        while block_size > 64 and p <= block_size:
            block_size //= 2
        while mt_pervar and block_size * t > 2048:                      # the markers' own constants are parked in LDS
            block_size //= 2
        # Rule T (jwas_sweep_params.section_solve, round 5): the dense 256-marker blocks of sampler I as triangular solves with
        # per-sweep section inverses (at most three traits, Float32 context, the plain single-pass schedule)
        section_solve = bool(dense and mt_big and t <= 3 and block_size == 256 and not double_precision and fast_blocks is False
                             and not independent_blocks)
        # Sparse priors: keep a second, larger block size resident and pick per sweep from the previous sweep's number
        # of effect changes (a chain quantity, so runs stay reproducible): 1024-marker blocks amortise the per-launch cost
        # once fewer than ~1.3 % of the markers change per sweep (measured crossover at 50k x 600k: 8 900 changes).
        # (single-trait only: the multi-trait samplers are bound by the serial phase, not by launches, and at 1024
        # markers x 3 traits the per-marker draws no longer fit LDS next to the staged Gram rows -- measured 10.0 vs 8.9 ms
        # per sweep at 20k x 100k x 3 traits)
        adaptive = (not dense) and t == 1 and block_size == 512 and p > 4 * 1024
        # (multi-trait chains that start dense: 256-marker blocks now, 512 once the chain has become sparse -- pick_block_size_mt)
        adaptive_mt = bool(dense and mt_big and block_size == 256 and p > 4 * 512 and 512 * t <= 2048)
        mt_1024 = adaptive_mt and mt_1024_allowed(t, p, mt_pervar)
        # (multi-trait chains that START sparse: 512-marker blocks, 1024 for the sweeps below 0.5 % turnover -- the same last level)
        adaptive_mts = bool((not dense) and t > 1 and block_size == 512 and mt_1024_allowed(t, p, mt_pervar)
                            and getattr(Mi, "annotations", False) is False)      # (marker-specific prior tables: their LDS copy does not fit at 1024 x 3)
        mt_1024 = mt_1024 or adaptive_mts

    if double_precision:
        # the Float64 device context (csrc/f64_path.hpp): dense storage; single-trait BayesA/B/C (+ RR-BLUP, BayesL through
        # them), BayesR, multi-trait BayesC sampler I; any fast_blocks partition with blocks of <= 1024 markers (uniform or
        # explicit starts), independent_blocks, residual weights; block size x traits <= 2048
        if stream or devres:
            raise NotImplementedError("double_precision=true needs dense host genotypes (the streaming backend is Float32 only, readgenotypes.jl:246-248)")
        if mega or mt_pervar or (t > 1 and mt_method != "MTBayesC"):
            raise NotImplementedError("double_precision=true runs the single-trait samplers and multi-trait BayesC sampler I on the device; "
                                      "sampler II, constraint=true and multi-trait BayesA/B stay on the reference in Float64 mode")
        if fast_blocks is not False:
            if explicit_partition is None:
                block_size = want                      # (the Float64 context runs uniform blocks of ANY size <= 1024 as they are)
        else:
            # two stream-ordered launches per block, no lookahead: big blocks amortise them (20k x 20k BayesC: 6.8 ms per sweep
            # at 128 markers per block, 2.1 ms at 512 -- scripts/f64_bench.py); the chain is the same whatever the size
            block_size = 512
            while block_size > 64 and p <= block_size:
                block_size //= 2
        if block_size * t > 2048:
            raise NotImplementedError(f"double_precision=true needs fast_blocks * traits <= 2048 on the device (got {block_size} x {t})")
        adaptive = False
        adaptive_mt = False
        adaptive_mts = False
        mt_1024 = False

    # grouped launches (engine.setup_groups): the adaptive policy's 1024-marker sweeps of a single-trait chain (dense or 2-bit packed
    # storage), when the chain is long enough to pay for the group cross-Grams; an explicitly partitioned / row-sharded run keeps
    # one block per launch
    group_m = 0
    if adaptive and not double_precision and explicit_partition is None and not independent_blocks and fast_blocks is False:
        # (blocks_per_launch: None = by the chain's length; 0 / 2 / 4 = the caller's choice -- a device option like block_size)
        group_m = grouped_blocks_for_chain(chain_length) if blocks_per_launch is None else int(blocks_per_launch)
        if not grouped_launch_size(method, t, False, 1024, group_m):
            group_m = 0
    pair_m = 0                                 # ... and ping-pong pairs on the 512-marker sweeps of a high-turnover chain (dense storage)
    if adaptive and not double_precision and explicit_partition is None and not independent_blocks and fast_blocks is False and not stream:
        pair_m = pingpong_pairs_for_chain(method, bool(Mi.estimatePi), chain_length) if blocks_per_launch is None else (2 if int(blocks_per_launch) else 0)
        if pair_m == 4 and 4 * p * 512 * 5 > 0.05 * 288e9:      # (pair + four cross-Grams of the 512-marker set: keep them a small part of the HBM)
            pair_m = 2
        if not grouped_launch_size(method, t, False, 512, pair_m):
            pair_m = 0

    from .engine import SectionSolvePolicy
    solve_policy = SectionSolvePolicy(section_solve, 4 * (p // 256))

    # ---- engine (the only engine shipped is the HIP one; there is no CPU fallback)
    own_engine = engine is None
    if own_engine:
        from .engine import HipEngine
        need = HipEngine.estimate_bytes(n, p, t, block_size, "stream" if stream else "dense") * (2 if double_precision else 1)
        if adaptive or adaptive_mt:
            need += 2 * 4 * (1024 if adaptive else 512) * p        # the second resident block size (Grams + cross-Grams)
        if mt_1024:
            need += 2 * 4 * 1024 * p                               # ... and the third one of a multi-trait chain
        if group_m:
            need += 4 * p * 1024 * (5 if group_m == 4 else 2)      # grouped launches: pair (4 blocks: the odd pairs + the fours) cross-Grams of the 1024-marker set
        if pair_m:
            need += 4 * p * 512 * (5 if pair_m == 4 else 2)       # ... and the pair (4 blocks: odd pairs + fours) cross-Grams of the 512-marker set
        if double_precision and independent_blocks:
            # Float64 independent blocks: one change list of 1024 entries per block (4 + 4 x 8 bytes per entry, whatever the
            # block size) and one partial-sum buffer per block (4 traits x row slices x block doubles)
            nb_ = -(-p // max(1, block_size))
            need += nb_ * (4 + 1024 * (4 + 4 * 8)) + 4 * 8 * (-(-n // 256)) * nb_ * block_size
        if outputEBV and not out_same:                             # Mi.output_genotypes: a second dense matrix (n_out x p)
            need += 4 * ((len(out_rows) + 255) // 256 * 256) * p
        engine = HipEngine(device, precision=64 if double_precision else 32)
        free = engine.device_info()["hbm_free"]
        if memory_guard != "off" and need > memory_guard_ratio * free:   # JWAS.jl:422-459 analogue for HBM
            msg = (f"marker path needs {need / 1e9:.2f} GB of HBM, more than {memory_guard_ratio:.2f} x free "
                   f"({free / 1e9:.2f} GB)")
            if memory_guard == "error":
                engine.close()
                raise MemoryError(msg)
            print("WARNING: " + msg)
    if devres:
        pass                                   # already resident
    elif stream:
        engine.load_jgb2(Mi.stream_backend["prefix"])          # payload stays 2-bit packed in HBM
    else:
        engine.load_dense(np.asfortranarray(X, dtype=ftype))      # after alignment (tools4genotypes.jl:310-321); Float64 on request (JWAS.jl:353)
    X_out_host = None
    if outputEBV and not out_same:             # Mi.output_genotypes = Z_out * genotypes (tools4genotypes.jl:290-296)
        if double_precision:                   # (the Float64 context has no second resident matrix: the EBV product runs on the host)
            X_out_host = np.asarray(Mi.genotypes[out_rows, :], dtype=np.float64)
        else:
            engine.load_output_dense(np.asfortranarray(Mi.genotypes[out_rows, :]))
    # A device-resident engine may come from an earlier run: its weights, Grams and block partition must be THIS run's.
    # set_weights(None) restores unit weights (and drops the resident Grams, which were X_b'R^-1 X_b); an explicit
    # partition left by the previous run is rebuilt as uniform blocks below.
    if invw is not None or getattr(engine, "_weighted", False):
        engine.set_weights(invw)               # x'R^-1 x, X_b'R^-1 X_b, X_b'R^-1 r on the device (GibbsMats with Rinv)
    if explicit_partition is not None:
        engine.setup_blocks_explicit(explicit_partition, gram_mode)
    elif devres and invw is None and engine.block_size and getattr(engine, "_explicit_starts", None) is None:
        resident = set(engine.resident_block_sizes())
        if block_size not in resident:
            engine.add_block_size(block_size, gram_mode)
        if (adaptive or adaptive_mt) and (1024 if adaptive else 512) not in resident:
            engine.add_block_size(1024 if adaptive else 512, gram_mode)
        if mt_1024 and 1024 not in resident:
            engine.add_block_size(1024, gram_mode)
        engine.select_block_size(block_size)
    else:
        engine.setup_blocks(block_size, gram_mode)
        if adaptive or adaptive_mt:
            engine.add_block_size(1024 if adaptive else 512, gram_mode)
        if mt_1024:
            engine.add_block_size(1024, gram_mode)
    if group_m:            # grouped launches on the large block size of the adaptive policy (setup work, once)
        if engine.blocks_per_launch(1024) != group_m:
            cur_bs = engine.block_size
            engine.select_block_size(1024)
            try:
                engine.setup_groups(group_m, gram_mode)
            except Exception as ex:      # noqa: BLE001  (no HBM left for the group cross-Grams on an injected engine, an unsupported storage, ...)
                print(f"NOTICE: grouped launches not set up ({ex}); the chain runs one block per launch.")
                group_m = 0
            engine.select_block_size(cur_bs)
    if pair_m and engine.blocks_per_launch(512) != pair_m:
        cur_bs = engine.block_size
        engine.select_block_size(512)
        try:
            engine.setup_groups(pair_m, gram_mode)
        except Exception as ex:      # noqa: BLE001
            print(f"NOTICE: ping-pong pairs not set up ({ex}); the 512-marker sweeps run one block per launch.")
            pair_m = 0
        engine.select_block_size(cur_bs)
    engine.init_state(mt_method if t > 1 else method, t)

    # ---- fixed effects
    Xf, labels = _design(model, ph, idcol)
    q = [x.shape[1] for x in Xf]
    sol = np.zeros(sum(q))
    off = np.cumsum([0] + q)

    # ---- starting state (MCMC_BayesianAlphabet.jl:85-147)
    alpha0 = np.zeros((t, p), dtype=ftype)
    if Mi.alpha is not False:
        alpha0[:] = np.asarray(Mi.alpha, dtype=ftype).reshape(t, p)
    for k in range(t):
        engine.set_state(k, alpha=alpha0[k], beta=alpha0[k],
                         delta=np.ones(p, dtype=np.int32 if method == "BayesR" else ftype))
        engine.set_residual(Y[k], k)           # sol = 0
        if alpha0[k].any():
            engine.sub_xalpha(k)

    vare = ftype(R.val) if t == 1 else np.asarray(R.val, dtype=ftype)
    Gval = ftype(Mi.G.val) if t == 1 else np.asarray(Mi.G.val, dtype=ftype)
    if lasso:                                                           # MCMC_BayesianAlphabet.jl:70-81
        Gval = ftype(Gval / 8)
        Mi.G.scale = Mi.G.scale / 8
        gamma_l = rng.gamma(1.0, 8.0, size=p)
        Gvec = (np.float64(Gval) * gamma_l).astype(ftype)
    elif mt_pervar:
        Gmat = np.tile(np.asarray(Gval, dtype=ftype), (p, 1, 1))   # MCMC_BayesianAlphabet.jl:67-69 (fill(G, nMarkers))
    elif method == "BayesB":
        Gvec = np.full(p, Gval, dtype=ftype)                       # MCMC_BayesianAlphabet.jl:67-69
    pervar = method == "BayesB" and not lasso                           # per-marker variances, no common variance to report
    if t == 1 and method in ("BayesC", "BayesB") and np.ndim(pi) == 0:
        pi = float(pi)
    if t > 1:
        lhs_blocks = [[Xf[k].T @ (w64[:, None] * Xf[l]) for l in range(t)] for k in range(t)]    # X'RiX, Ri = kron(R^-1, diag(w))
    else:
        lhs = Xf[0].T @ (w64[:, None] * Xf[0])                          # X'R^-1 X (build_MME.jl:339)

    # ---- accumulators and sample files (output.jl:320-437)
    run_sol, run_vare = _Running(sol), _Running(vare)
    run_varg = _Running(Gval) if not pervar else None
    run_pi = _Running(np.atleast_1d(np.asarray(pi_t if mega else pi, dtype=np.float64))) if Mi.estimatePi else None
    ebv_run = [_Running(np.zeros(len(out_ids))) for _ in range(t)] if outputEBV else None
    run_scale = _Running(np.atleast_1d(np.float64(Mi.G.scale))) if (Mi.G.estimate_scale and t == 1) else None
    name = Mi.name
    files = {}

    def _open(key, header):
        fh = open(os.path.join(output_folder, f"MCMC_samples_{key}.txt"), "w")
        fh.write(",".join(header) + "\n")
        files[key] = fh

    rnames = [f"{a}_{b}" for a in model.lhsVec for b in model.lhsVec]
    _open("residual_variance", rnames if t > 1 else [model.lhsVec[0]])
    if not pervar:
        _open(f"marker_effects_variances_{name}", rnames if t > 1 else ["1"])
    if Mi.estimatePi and (np.size(pi) <= 20000 or output_samples_for_all_parameters):   # (marker-level pi: p values per sample)
        npi = t if mega else np.size(pi)
        _open(f"pi_{name}", [f"pi{i + 1}" for i in range(npi)] if npi > 1 else ["pi"])
    heritability = bool(outputEBV and output_heritability)              # output.jl:358-362,426-432
    if heritability:
        _open("genetic_variance", rnames if t > 1 else [model.lhsVec[0]])
        _open("heritability", list(model.lhsVec))
        h2_samples, gv_samples = [], []
    term_cols = {}                                                      # outputMCMCsamples (output.jl:76-95,443-460)
    for tr, trm in getattr(model, "outputSamplesVec", []):
        k = model.lhsVec.index(tr)
        cols = [off[k] + i for i, (_, eff, _) in enumerate(labels[k]) if eff == trm]
        if cols:
            term_cols[f"{tr}.{trm}"] = cols
            _open(f"{tr}.{trm}", [f"{tr}:{trm}:{labels[k][c - off[k]][2]}" for c in cols])
    write_marker_samples = output_samples_for_all_parameters or p <= 20000
    if write_marker_samples:
        for k, tr in enumerate(model.lhsVec):
            _open(f"marker_effects_{name}_{tr}", Mi.markerID)
    # every saved sample also goes to a binary file of sparse records -- the nonzero effects compacted on the device
    # (samples.py); the text row of p values (output.jl:467) only for small p or on request
    from .samples import MarkerSampleWriter
    bin_writers = [MarkerSampleWriter(os.path.join(output_folder, f"MCMC_samples_marker_effects_{name}_{tr}.bin"), Mi.markerID)
                   for tr in model.lhsVec]

    t_sweep = 0.0
    iter_end = []                                # perf_counter at the end of every iteration (each ends synchronised with the device)
    # The host step between two sweeps is a handful of O(n) vector operations.  A threaded BLAS runs them on every core and
    # its workers keep spinning afterwards: in a container with a CPU quota that burns the quota and stalls the NEXT device
    # call until the scheduler's next 100 ms period (measured: 78-100 ms per iteration instead of 21 at 50k x 600k).
    # JWAS_HOST_BLAS_THREADS overrides (0 = leave the BLAS alone).
    _blas_limit = None
    try:
        nthr = int(os.environ.get("JWAS_HOST_BLAS_THREADS", "1"))
        if nthr > 0:
            from threadpoolctl import threadpool_limits
            _blas_limit = threadpool_limits(limits=nthr, user_api="blas")
    except Exception:                            # threadpoolctl missing: nothing to limit with
        _blas_limit = None
    t0 = time.time()
    # ================================ the chain =================================================
    try:
        for it in range(1, chain_length + 1):
            # 1. location parameters (host)
            if sum(q):
                if t == 1:
                    r = engine.get_residual(0).astype(np.float64)
                    r += Xf[0] @ sol
                    rhs = Xf[0].T @ (w64 * r)                               # MCMC_BayesianAlphabet.jl:211
                    _gibbs(lhs, sol, rhs, rng, float(vare))
                    r -= Xf[0] @ sol
                    engine.set_residual(r.astype(ftype), 0)
                else:
                    R0 = np.asarray(vare, dtype=np.float64)
                    Rinv = np.linalg.inv(R0)
                    res = [engine.get_residual(k).astype(np.float64) for k in range(t)]
                    Ri_rows = _impute_missing_residuals(res, observed, R0, rng) if has_missing else None
                    if has_missing and (it == 1 or not R.estimate_variance):
                        # Residuals of the missing records are imputed from the observed ones every iteration
                        # (sampleMissingResiduals, residual.jl:52-73).  The location-parameter equations weight every record
                        # with the inverse of the OBSERVED block of R only (mkRi / getRi, residual.jl:2-44) -- but only until
                        # the first residual-variance draw: the reference then replaces Ri by kron(inv(R), diag(invweights))
                        # (MCMC_BayesianAlphabet.jl:357-361) and runs plain data augmentation on the imputed residuals.
                        if invw is not None:
                            Ri_rows = Ri_rows * w64[:, None, None]
                        rr = [res[k] + Xf[k] @ sol[off[k]:off[k + 1]] for k in range(t)]
                        A = np.block([[Xf[k].T @ (Ri_rows[:, k, l][:, None] * Xf[l]) for l in range(t)] for k in range(t)])
                        b = np.concatenate([Xf[k].T @ sum(Ri_rows[:, k, l] * rr[l] for l in range(t)) for k in range(t)])
                    else:
                        rr = [res[k] + Xf[k] @ sol[off[k]:off[k + 1]] for k in range(t)]
                        A = np.block([[Rinv[k, l] * lhs_blocks[k][l] for l in range(t)] for k in range(t)])
                        b = np.concatenate([Xf[k].T @ (w64 * sum(Rinv[k, l] * rr[l] for l in range(t))) for k in range(t)])
                    _gibbs(A, sol, b, rng, None)
                    for k in range(t):
                        engine.set_residual((rr[k] - Xf[k] @ sol[off[k]:off[k + 1]]).astype(ftype), k)

            elif t > 1 and has_missing:                                       # no location parameters: imputation only
                res = [engine.get_residual(k).astype(np.float64) for k in range(t)]
                _impute_missing_residuals(res, observed, np.asarray(vare, dtype=np.float64), rng)
                for k in range(t):
                    engine.set_residual(res[k].astype(ftype), k)

            # 2. marker effects (DEVICE)
            kw = dict(iteration=it, seed=seed_int, vare=vare, nreps=nreps)
            if independent_blocks:
                kw["independent_blocks"] = True
            if solve_policy.use(it) and engine.block_size == 256:
                kw["section_solve"] = True
            if mega:
                kw.update(var_effect=Gval, pi=pi_t)
                if mt_pervar:
                    kw["var_effect_matrix"] = Gmat
            elif t > 1:
                with np.errstate(divide="ignore"):
                    kw.update(var_effect=Gval, log_prior_states=np.log(ann.snp_pi if ann is not False else np.asarray(pi, dtype=np.float64)))
                if mt_pervar:
                    kw["var_effect_matrix"] = Gmat
            elif method == "BayesR":
                kw.update(var_effect=Gval, pi_classes=np.asarray(pi, dtype=np.float64))
                if ann is not False:
                    kw["pi_matrix"] = ann.snp_pi                            # per-marker class priors (BayesR.jl:62-66)
                if fast_blocks is not False:                                # bayesr_block_nreps (BayesR.jl:22-25)
                    kw["nreps"] = 1 if it <= burnin else nreps
            elif method == "BayesB":
                kw.update(var_effect=Gval, var_effect_vec=Gvec, pi=pi)
            elif np.ndim(pi) == 1:                                          # marker-level pi (bayesabc_pi_vector, BayesABC.jl:16-22)
                kw.update(var_effect=Gval, pi_vec=pi)
            else:
                kw.update(var_effect=Gval, pi=pi)
            if (group_m or pair_m) and engine.blocks_per_launch() >= 2:
                kw["group_launch"] = True
            st = engine.sweep(**kw)
            solve_policy.observe(it, engine, ran=bool(kw.get("section_solve")))
            t_sweep += st["sweep_ms"]
            if adaptive:
                engine.select_block_size(pick_block_size(st["n_events"], p, pairs=bool(pair_m)))
            elif adaptive_mt:
                engine.select_block_size(pick_block_size_mt(st["n_events"], p, allow_1024=mt_1024))
            elif adaptive_mts:
                engine.select_block_size(1024 if st["n_events"] < MT_1024_CHANGE_FRACTION * p else 512)

            # 3. pi (Pi.jl:7-42)
            if Mi.estimatePi:
                if mega:                                                    # MCMC_BayesianAlphabet.jl:300-301
                    pi_t = np.array([rng.beta(p - st["sum_delta"][k] + 1.0, st["sum_delta"][k] + 1.0) for k in range(t)])
                elif t > 1 and ann is not False:                            # annotation_updates.jl:353-361
                    pi = A_.update_bayesc_mt_tree_priors(ann, engine.get_state(0)[2], engine.get_state(1)[2], rng)
                elif t > 1:
                    pi = rng.dirichlet(st["state_counts"] + 1.0)
                elif ann is not False:                                      # update_marker_annotation_priors! (annotation_updates.jl:328-351)
                    dlt = engine.get_state(0)[2]
                    pi = A_.update_bayesr_nested_priors(ann, dlt, rng) if method == "BayesR" else A_.update_bayesc_binary_priors(ann, dlt, rng)
                elif method == "BayesR":
                    pi = rng.dirichlet(st["class_counts"] + 1.0)
                else:
                    pi = float(rng.beta(p - st["sum_delta"][0] + 1.0, st["sum_delta"][0] + 1.0))

            # 4. marker effect variance (variance_components.jl:151-189), re-cast to Float32 (:323-325)
            if Mi.G.estimate_variance:
                if mega and mt_pervar:                                      # one scaled inverse chi-square per marker and trait
                    sc = np.diag(Gdf * np.diag(np.asarray(Mi.G.scale, dtype=np.float64)))      # (variance_components.jl:112-117,181-186)
                    if hasattr(engine, "sample_marker_covariances"):        # on the device, from the resident beta
                        engine.sample_marker_covariances(Gdf + 1.0, sc, seed=seed_int, iteration=it)
                        Gmat = None
                    else:
                        B = np.stack([engine.get_state(k)[1] for k in range(t)], axis=1).astype(np.float64)
                        Gmat = np.zeros((p, t, t), dtype=ftype)
                        for k in range(t):
                            Gmat[:, k, k] = (B[:, k] ** 2 + sc[k, k]) / rng.chisquare(Gdf + 1.0, size=p)
                elif mega:                                                  # diagonal only (variance_components.jl:104-109)
                    Gval = np.diag([(st["beta_ss"][k, k] + Gdf * Mi.G.scale[k, k]) / rng.chisquare(p + Gdf) for k in range(t)]).astype(ftype)
                elif mt_pervar:                                             # variance_components.jl:181-186: one draw per marker,
                    if hasattr(engine, "sample_marker_covariances"):        # IW(df + 1, scale + b_j b_j') -- on the device, from the
                        # resident beta (jwas_hip_sample_marker_covariances: Bartlett on the counter RNG); the next sweep uses them in place
                        engine.sample_marker_covariances(Gdf + 1.0, np.asarray(Mi.G.scale, dtype=np.float64), seed=seed_int, iteration=it)
                        Gmat = None
                    else:
                        B = np.stack([engine.get_state(k)[1] for k in range(t)], axis=1).astype(np.float64)
                        Gmat = _inverse_wishart_batch(rng, Gdf + 1.0, np.asarray(Mi.G.scale, dtype=np.float64)[None] + B[:, :, None] * B[:, None, :]).astype(ftype)
                elif t > 1:
                    from scipy.stats import invwishart
                    S = np.asarray(Mi.G.scale, dtype=np.float64) + st["beta_ss"]
                    Gval = np.asarray(invwishart.rvs(df=Gdf + p, scale=(S + S.T) / 2, random_state=rng), dtype=ftype).reshape(t, t)
                elif method == "BayesR":
                    Gval = ftype((st["bayesr_ssq"] + Gdf * Mi.G.scale) / rng.chisquare(st["bayesr_nnz"] + Gdf))
                elif lasso:                                                 # variance_components.jl:152-166,191-203
                    a64 = engine.get_state(0)[0].astype(np.float64)
                    Gval = ftype((np.dot(a64 / gamma_l, a64) + Gdf * Mi.G.scale) / rng.chisquare(p + Gdf))
                    Q = a64 * a64 / np.float64(Gval)
                    cand = 1.0 / rng.gamma(0.5, 4.0, size=p)
                    with np.errstate(over="ignore"):
                        accept = rng.random(p) < np.exp(Q / 4.0 * (2.0 / gamma_l - cand))
                    gamma_l[accept] = 2.0 / cand[accept]
                    Gvec = (np.float64(Gval) * gamma_l).astype(ftype)
                elif method == "BayesB":
                    beta = engine.get_state(0)[1].astype(np.float64)
                    Gvec = ((beta * beta + Gdf * Mi.G.scale) / rng.chisquare(1.0 + Gdf, size=p)).astype(ftype)
                else:
                    Gval = ftype((ftype(st["alpha_ss"][0, 0]) + Gdf * Mi.G.scale) / rng.chisquare(st["sum_delta"][0] + Gdf))

            # 4b. scale of the marker-effect variance prior (MCMC_BayesianAlphabet.jl:328-336; single trait only there too)
            if Mi.G.estimate_scale and t == 1:
                gv = Gvec.astype(np.float64) if pervar else np.atleast_1d(np.float64(Gval))
                Mi.G.scale = float(rng.gamma(gv.size * Gdf / 2 + 1, 1.0 / (np.sum(Gdf / (2 * gv)) + 1)))

            # 5. residual variance (variance_components.jl:60-66,82-112), re-cast to Float32 (:368-370)
            if R.estimate_variance:
                if t > 1 and R.constraint:                                  # variance_components.jl:104-109
                    vare = np.diag([(st["resid_ss"][k, k] + Rdf * R.scale[k, k]) / rng.chisquare(n + Rdf) for k in range(t)]).astype(ftype)
                elif t > 1:
                    from scipy.stats import invwishart
                    S = np.asarray(R.scale, dtype=np.float64) + st["resid_ss"]
                    vare = np.asarray(invwishart.rvs(df=Rdf + n, scale=(S + S.T) / 2, random_state=rng), dtype=ftype).reshape(t, t)
                else:
                    vare = ftype((ftype(st["resid_ss"][0, 0]) + Rdf * R.scale) / rng.chisquare(n + Rdf))

            # 6. save (MCMC_BayesianAlphabet.jl:399-413, output.jl:443-604)
            if it > burnin and (it - burnin) % output_samples_frequency == 0:
                k = (it - burnin) / output_samples_frequency
                run_sol.add(sol, k)
                run_vare.add(vare, k)
                if run_varg is not None:
                    run_varg.add(Gval, k)
                if run_pi is not None:
                    run_pi.add(np.atleast_1d(pi_t if mega else pi), k)
                engine.accumulate(k)
                if run_scale is not None:
                    run_scale.add(np.atleast_1d(np.float64(Mi.G.scale)), k)
                if ann is not False:
                    A_.accumulate(ann, k)
                for key_, cols in term_cols.items():
                    files[key_].write(",".join(repr(float(sol[c])) for c in cols) + "\n")
                files["residual_variance"].write(",".join(repr(float(v)) for v in np.atleast_1d(vare).ravel()) + "\n")
                if not pervar:
                    files[f"marker_effects_variances_{name}"].write(",".join(repr(float(v)) for v in np.atleast_1d(Gval).ravel()) + "\n")
                if Mi.estimatePi and f"pi_{name}" in files:
                    files[f"pi_{name}"].write(",".join(repr(float(v)) for v in np.atleast_1d(pi_t if mega else pi)) + "\n")
                for kk, tr in enumerate(model.lhsVec):
                    if hasattr(engine, "alpha_sparse") and not double_precision:
                        si, sv = engine.alpha_sparse(kk)              # (idx, val) compacted on the device
                    else:
                        a_ = engine.get_state(kk)[0]
                        si = np.flatnonzero(a_).astype(np.int32); sv = a_[si]
                    bin_writers[kk].append(si, sv)
                    if write_marker_samples:
                        a = np.zeros(p, dtype=ftype)
                        a[si] = sv
                        fh = files[f"marker_effects_{name}_{tr}"]
                        a.tofile(fh, sep=",", format="%.9g")          # text at C speed; 9 significant digits round-trip Float32
                        fh.write("\n")
                if outputEBV:
                    if X_out_host is not None:
                        ebvs = [(X_out_host @ engine.get_state(kk)[0].astype(np.float64)).astype(ftype) for kk in range(t)]
                    else:
                        ebvs = [engine.mul_alpha(kk) if out_same else engine.mul_alpha_output(kk) for kk in range(t)]   # getEBV, output.jl:281-306
                    for kk in range(t):
                        ebv_run[kk].add(ebvs[kk], k)
                    if heritability:                                        # output.jl:498-512
                        E = np.stack(ebvs, axis=1).astype(np.float64)
                        gv = np.atleast_2d(np.cov(E, rowvar=False))
                        if t > 1 and Mi.G.constraint:
                            gv = np.diag(np.diag(gv))
                        vr = np.atleast_2d(np.asarray(vare, dtype=np.float64))
                        h2 = np.diag(gv) / (np.diag(gv) + np.diag(vr))
                        gv_samples.append(gv.ravel()); h2_samples.append(h2)
                        files["genetic_variance"].write(",".join(repr(float(v)) for v in gv.ravel()) + "\n")
                        files["heritability"].write(",".join(repr(float(v)) for v in h2) + "\n")
            iter_end.append(time.perf_counter())
            if it % printout_frequency == 0 and it > burnin:
                print(f"\nPosterior means at iteration: {it}")
                print(f"Residual variance: {np.round(run_vare.mean, 6)}")
    finally:                                     # also on an exception inside the chain: give the BLAS its threads back, close the files
        wall = time.time() - t0
        if _blas_limit is not None:
            _blas_limit.restore_original_limits()
        for fh in files.values():
            fh.close()
        for w_ in bin_writers:
            w_.close()


    # ---- results (output.jl:108-212)
    out = {}
    rows_lp = []
    for k in range(t):
        for i, (tr, eff, lev) in enumerate(labels[k]):
            rows_lp.append((tr, eff, lev, run_sol.mean[off[k] + i], run_sol.sd()[off[k] + i]))
    out["location parameters"] = pd.DataFrame(rows_lp, columns=["Trait", "Effect", "Level", "Estimate", "SD"])
    cov = rnames if t > 1 else [model.lhsVec[0]]
    out["residual variance"] = pd.DataFrame({"Covariance": cov, "Estimate": np.atleast_1d(run_vare.mean).ravel(),
                                             "SD": np.atleast_1d(run_vare.sd()).ravel()})
    frames = []
    for k, tr in enumerate(model.lhsVec):
        ma, ma2, md = engine.posterior(k)
        sd = np.sqrt(np.abs(ma2.astype(np.float64) - ma.astype(np.float64) ** 2))
        frames.append(pd.DataFrame({"Trait": tr, "Marker_ID": Mi.markerID, "Estimate": ma, "SD": sd, "Model_Frequency": md}))
    out[f"marker effects {name}"] = pd.concat(frames, ignore_index=True)
    if run_varg is not None:
        out[f"marker effects variance {name}"] = pd.DataFrame({"Covariance": cov, "Estimate": np.atleast_1d(run_varg.mean).ravel(),
                                                               "SD": np.atleast_1d(run_varg.sd()).ravel()})
    if run_pi is not None:
        if mega:
            lab = list(model.lhsVec)
        elif t > 1:
            lab = ["".join(str((s >> k) & 1) for k in range(t)) for s in range(1 << t)]
        elif method == "BayesR":
            lab = ["class1", "class2", "class3", "class4"]
        elif np.size(run_pi.mean) > 1:
            lab = list(Mi.markerID)                                      # annotated BayesC: marker-level pi
        else:
            lab = ["π"]
        out[f"pi_{name}"] = pd.DataFrame({"π": lab, "Estimate": run_pi.mean, "SD": run_pi.sd()})
    if ann is not False:
        out[f"annotation coefficients {name}"] = A_.coefficients_table(ann, method)
    if run_scale is not None:                                            # output.jl:148-150
        out[f"ScaleEffectVar{name}"] = pd.DataFrame({"Covariance": [model.lhsVec[0]], "Estimate": run_scale.mean, "SD": run_scale.sd()})
    if outputEBV:
        for k, tr in enumerate(model.lhsVec):
            m = ebv_run[k].mean
            out[f"EBV_{tr}"] = pd.DataFrame({"ID": out_ids, "EBV": m, "PEV": np.abs(ebv_run[k].mean2 - m ** 2)})
    if heritability and gv_samples:                                      # output.jl:196-209 (mean and std of the samples)
        for key, samples, names in (("genetic_variance", np.array(gv_samples), cov), ("heritability", np.array(h2_samples), list(model.lhsVec))):
            out[key] = pd.DataFrame({"Covariance": names, "Estimate": samples.mean(axis=0),
                                     "SD": samples.std(axis=0, ddof=1) if len(samples) > 1 else np.full(samples.shape[1], np.nan)})
    for key, tab in out.items():                                         # JWAS.jl:480-482
        tab.to_csv(os.path.join(output_folder, key.replace(" ", "_") + ".txt"), index=False)
    out["_timing"] = {"wall_s": wall, "device_sweep_ms_total": t_sweep, "iterations": chain_length,
                      "block_size": block_size, "n": n, "p": p, "iteration_end_s": iter_end,
                      "block_starts": (np.asarray(engine.block_starts(), dtype=np.int64) + 1).tolist() if fast_blocks is not False else None,
                      "block_repetitions": int(nreps)}
    if own_engine and not devres:
        engine.close()
    return out
