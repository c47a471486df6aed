"""Host-side mirror of the reference's user API for the marker path:
get_genotypes() / build_model() / set_covariate() / runMCMC().

Same names, argument meaning and error behaviour as reworkhow/JWAS.jl v2.3.6 (files under
/root/reference/src/1.JWAS/src/):
    get_genotypes   markers/readgenotypes.jl:213-448     (dense branch)
    build_model     build_MME.jl:42-156
    set_covariate   build_MME.jl:158-181
    runMCMC         JWAS.jl:161-511  -> MCMC/MCMC_BayesianAlphabet.jl
Everything that is NOT the marker sweep stays here on the host (numpy): data ingestion and QC,
model parsing, fixed-effect Gibbs, variance-component and pi draws, output tables.  Model families
outside the hot path (pedigree / random terms, GBLUP, RR-BLUP, BayesL, categorical / censored traits,
SEM, RRM, single-step pre-processing, marker annotations) are rejected with an explicit error:
they stay on the reference.
"""
import inspect
import os

import numpy as np

from .mcmc import run_chain
from .gwas import GWAS  # noqa: F401  (src/3.GWAS/src/GWAS.jl)

# RR-BLUP: every marker is in the model with one common effect variance -- the same full conditionals as BayesC with
# pi = 0 fixed (BayesL! with gammaArray = [1.0], BayesC0L.jl:20-47, vs bayesabc_update_marker! with probDelta1 = 1;
# variance update with nloci = nMarkers, variance_components.jl:160-162), so it runs on the device's BayesC path
# BayesL (single trait): BayesL!'s update is the device's BayesB update with pi = 0 and var_j = G*gamma_j; the gamma_j
# Metropolis-Hastings step stays on the host (mcmc.py)
SUPPORTED_METHODS = ("BayesA", "BayesB", "BayesC", "BayesR", "RR-BLUP", "BayesL")
BAYESR_DEFAULT_PI = np.array([0.95, 0.03, 0.015, 0.005])      # tools4genotypes.jl:375-377
BAYESR_GAMMA = np.array([0.0, 0.01, 0.1, 1.0])                # JWAS.jl:12


class Variance:
    """types.jl:56-64"""

    def __init__(self, val, df, scale, estimate_variance=True, estimate_scale=False, constraint=False):
        self.val, self.df, self.scale = val, df, scale
        self.estimate_variance, self.estimate_scale, self.constraint = estimate_variance, estimate_scale, constraint

    def __repr__(self):
        return f"Variance(val={self.val}, df={self.df}, scale={self.scale}, estimate_variance={self.estimate_variance})"


class Genotypes:
    """types.jl:98-165 -- the fields the marker path reads."""

    def __init__(self, obsID, markerID, nObs, nMarkers, alleleFreq, sum2pq, centered, genotypes):
        self.name = False
        self.trait_names = False
        self.obsID, self.markerID = list(obsID), list(markerID)
        self.nObs, self.nMarkers = int(nObs), int(nMarkers)
        self.alleleFreq, self.sum2pq, self.centered = alleleFreq, float(sum2pq), bool(centered)
        self.genotypes = genotypes
        self.ntraits = False
        self.genetic_variance = Variance(False, False, False)
        self.G = Variance(False, False, False)
        self.method = False
        self.estimatePi = True
        self.pi = 0.0
        self.alpha = False                      # starting values
        self.storage_mode = "dense"            # "dense" | "stream" (types.jl:149-150)
        self.stream_backend = None             # metadata of the 2-bit packed backend (streaming.load_streaming_backend)
        self.multi_trait_sampler = "I"
        self.annotations = False               # annotations.MarkerAnnotations (types.jl:167-215)


def _is_false(x):
    return x is False or x is None


def get_genotypes(file, G=False, *, method="BayesC", Pi=0.0, estimatePi=True,
                  G_is_marker_variance=False, df=4.0, estimate_variance=True, estimate_scale=False,
                  constraint=False, separator=",", header=True, double_precision=False,
                  quality_control=True, MAF=0.01, missing_value=9.0, center=True, starting_value=False,
                  annotations=False, multi_trait_sampler="I", storage="dense"):
    """readgenotypes.jl:213-448.  `file`: path of a delimited text file (first column = individual
    IDs), a pandas DataFrame (first column IDs) or a 2-D array."""
    if multi_trait_sampler not in ("auto", "I", "II"):
        raise ValueError("multi_trait_sampler must be one of :auto, :I, or :II.")            # :229-231
    if storage not in ("dense", "stream"):
        raise ValueError("storage must be :dense or :stream.")                                # :232-234
    if storage == "stream":                                                                   # :236-295
        if not isinstance(file, str):
            raise ValueError("storage=:stream requires a file path or prefix produced by prepare_streaming_genotypes.")
        if double_precision:
            raise ValueError("storage=:stream MVP supports Float32 only (double_precision=false).")
        if annotations is not False:
            raise NotImplementedError("marker annotations stay on the reference path")
        if method not in SUPPORTED_METHODS:
            raise NotImplementedError(f"method {method} is not on the device path (supported: {SUPPORTED_METHODS})")
        from .streaming import load_streaming_backend
        backend = load_streaming_backend(file)
        if center != backend["centered"]:
            print(f"The argument center={center} is ignored for storage=:stream. Backend metadata "
                  f"centered={backend['centered']} is used.")
        g = Genotypes(backend["obsID"], backend["markerID"], backend["nObs"], backend["nMarkers"],
                      backend["allele_freq"].astype(np.float32), backend["sum2pq"], backend["centered"],
                      np.zeros((backend["nObs"], 0), dtype=np.float32))
        g.storage_mode, g.stream_backend = "stream", backend
        g.G = Variance(G if G_is_marker_variance else False, df, False, estimate_variance, estimate_scale, constraint)
        g.genetic_variance = Variance(False if G_is_marker_variance else G, df, False, estimate_variance, estimate_scale, constraint)
        g.method, g.estimatePi, g.pi = method, estimatePi, Pi
        g.multi_trait_sampler = multi_trait_sampler
        print("Genotype informatin:")
        print(f"#markers: {backend['nMarkers']}; #individuals: {backend['nObs']} (storage=:stream)")
        if not _is_false(starting_value):
            sv = np.asarray(starting_value, dtype=np.float32).reshape(-1)
            if sv.size % backend["nMarkers"] != 0:
                raise ValueError("length of starting values is wrong.")
            g.alpha = sv
        return g
    if method not in SUPPORTED_METHODS:
        raise NotImplementedError(f"method {method} is not on the device path (supported: {SUPPORTED_METHODS}); "
                                  "GBLUP / RR-BLUP / BayesL stay on the reference")

    data_type = np.float64 if double_precision else np.float32           # readgenotypes.jl:298: Float64 genotypes on request
    try:
        import pandas as pd
    except ImportError:                                              # pragma: no cover
        pd = None
    if isinstance(file, str):                                                                 # :300-327
        with open(file) as fh:
            row1 = [t.strip().strip('"') for t in fh.readline().rstrip("\r\n").split(separator) if t != ""]
        ncol = len(row1)
        markerID = [str(t) for t in row1[1:]] if header else [str(i + 1) for i in range(ncol - 1)]
        from collections import defaultdict
        # marker columns are parsed straight to Float32 (the reference's default precision): half the transient memory
        tab = pd.read_csv(file, sep=separator, header=None, skiprows=1 if header else 0,
                          dtype=defaultdict(lambda: data_type, {0: str}))
        obsID = [str(v) for v in tab.iloc[:, 0]]
        genotypes = np.asfortranarray(tab.iloc[:, 1:].to_numpy(dtype=data_type))
    elif pd is not None and isinstance(file, pd.DataFrame):                                   # :328-338
        markerID = [str(c) for c in file.columns[1:]] if header else [str(i + 1) for i in range(file.shape[1] - 1)]
        obsID = [str(v) for v in file.iloc[:, 0]]
        genotypes = np.asfortranarray(file.iloc[:, 1:].to_numpy(dtype=data_type))
    elif isinstance(file, np.ndarray) and file.ndim == 2:                                      # :339-345
        markerID = [str(i + 1) for i in range(file.shape[1])]
        obsID = [str(i + 1) for i in range(file.shape[0])]
        genotypes = np.array(file, dtype=data_type, order="F")
    else:
        raise TypeError("The data type is not supported.")                                    # :346-348

    nObs, nMarkers = genotypes.shape
    from . import annotations as _ann
    annotation_matrix = _ann.validate_annotations_input(annotations, nMarkers, method)       # :56-70, one row per RAW marker
    if annotation_matrix is not False and estimatePi is False:                                # :152-158
        print(f"estimatePi=false is ignored when annotations are provided; Annotated {method} requires estimatePi=true.")
        estimatePi = True
    if quality_control:                                                                       # :372-382
        mv = data_type(missing_value)
        for j in range(nMarkers):
            col = genotypes[:, j]
            miss = col == mv
            if miss.any():
                col[miss] = col[~miss].mean(dtype=data_type)
    if center:                                                                                # :384
        markerMeans = genotypes.mean(axis=0, dtype=data_type).astype(data_type)
        genotypes -= markerMeans[None, :]
    else:
        markerMeans = genotypes.mean(axis=0, dtype=data_type).astype(data_type)
    p = (markerMeans / data_type(2.0)).astype(data_type)                                      # :385
    if quality_control:                                                                       # :388-399
        select1 = (MAF < p) & (p < 1 - MAF)
        select2 = genotypes.var(axis=0, ddof=1) != 0
        select = select1 & select2
        genotypes = np.asfortranarray(genotypes[:, select])
        p = p[select]
        markerID = [m for m, s in zip(markerID, select) if s]
        if annotation_matrix is not False:
            annotation_matrix = annotation_matrix[select, :]
        print(f"{int((~select).sum())} loci which are fixed or have minor allele frequency < {MAF} are removed.")
    nObs, nMarkers = genotypes.shape
    sum2pq = float((2.0 * p.astype(data_type) * (1.0 - p.astype(data_type))).sum(dtype=data_type))   # :401

    g = Genotypes(obsID, markerID, nObs, nMarkers, p, sum2pq, center, genotypes)
    g.G = Variance(G if G_is_marker_variance else False, df, False, estimate_variance, estimate_scale, constraint)   # :424
    g.genetic_variance = Variance(False if G_is_marker_variance else G, df, False, estimate_variance, estimate_scale, constraint)
    g.method, g.estimatePi, g.pi = method, estimatePi, Pi
    g.multi_trait_sampler = multi_trait_sampler
    if annotation_matrix is not False:                                                        # :111-150
        if method == "BayesC" and not isinstance(Pi, dict):
            if np.ndim(Pi) == 1:
                if len(Pi) != nMarkers:
                    raise ValueError(f"Annotated BayesC starting Pi vector length {len(Pi)} must match the number of markers ({nMarkers}).")
                g.pi = np.array(Pi, dtype=np.float64)
            else:
                g.pi = np.full(nMarkers, float(Pi))
        g.annotations = _ann.build_marker_annotations(annotation_matrix, method, Pi)
    print("Genotype informatin:")
    print(f"#markers: {nMarkers}; #individuals: {nObs}")
    if not _is_false(starting_value):                                                         # :438-446
        sv = np.asarray(starting_value, dtype=data_type).reshape(-1)
        if sv.size % nMarkers != 0:
            raise ValueError("length of starting values is wrong.")
        g.alpha = sv
    return g


def device_genotypes(engine, G=False, *, method="BayesC", Pi=0.0, estimatePi=True, G_is_marker_variance=False, df=4.0,
                     estimate_variance=True, estimate_scale=False, constraint=False, multi_trait_sampler="I",
                     obsID=None, markerID=None, centered=True, alleleFreq=None, sum2pq=None):
    """A Genotypes object over a matrix that is ALREADY resident on the GPU (loaded or generated through `engine`, a
    HipEngine) -- the device analogue of storage=:stream, where Genotypes carries a backend handle instead of the matrix
    (types.jl:149-150, readgenotypes.jl:236-295).  Like that mode: phenotype IDs must match the genotype IDs exactly and
    in order, Float32 only.  sum2pq / alleleFreq: pass them when they are known (impute_genotypes does; the reference
    computes sum(2p(1-p)) from the column means before centring, readgenotypes.jl:385-402); otherwise they are derived from
    the device's x'x, which is exact only for CENTRED 0/1/2 genotypes with unit residual weights (x'x / n = 2pq) -- an
    uncentred or weighted resident matrix without explicit values is an error, not a silently wrong variance conversion.
    Block configurations already resident on the engine are used as they are."""
    if method not in SUPPORTED_METHODS:
        raise NotImplementedError(f"method {method} is not on the device path (supported: {SUPPORTED_METHODS})")
    if multi_trait_sampler not in ("auto", "I", "II"):
        raise ValueError("multi_trait_sampler must be one of :auto, :I, or :II.")
    n, p = int(engine.n), int(engine.p)
    if n <= 0 or p <= 0:
        raise ValueError("the engine holds no genotype matrix")
    if engine.block_size == 0:
        engine.setup_blocks(512 if p > 512 else 64, "mfma")
    if sum2pq is None or alleleFreq is None:
        if not centered:
            raise ValueError("device_genotypes(centered=False): x'x / n is 2pq + 4p^2 for an uncentred matrix; pass alleleFreq "
                             "and sum2pq explicitly")
        if getattr(engine, "_weighted", False):
            raise ValueError("device_genotypes: the engine holds residual weights (x'x is x'R^-1 x); pass alleleFreq and "
                             "sum2pq explicitly")
        xpx = engine.xpx().astype(np.float64)
        if sum2pq is None:
            sum2pq = float(xpx.sum() / n)
        if alleleFreq is None:      # allele frequency from 2pq = x'x / n (the smaller root); only marker-level-pi priors read it
            alleleFreq = 0.5 * (1.0 - np.sqrt(np.clip(1.0 - 2.0 * xpx / n, 0.0, 1.0)))
    af = np.asarray(alleleFreq, dtype=np.float32).reshape(-1)
    if af.shape != (p,):
        raise ValueError(f"alleleFreq must have one entry per marker ({p})")
    sum2pq = float(sum2pq)
    g = Genotypes(obsID if obsID is not None else [str(i + 1) for i in range(n)],
                  markerID if markerID is not None else [str(j + 1) for j in range(p)], n, p, af, sum2pq, centered,
                  np.zeros((n, 0), dtype=np.float32))
    g.storage_mode, g.device_backend = "device", engine
    g.G = Variance(G if G_is_marker_variance else False, df, False, estimate_variance, estimate_scale, constraint)
    g.genetic_variance = Variance(False if G_is_marker_variance else G, df, False, estimate_variance, estimate_scale, constraint)
    g.method, g.estimatePi, g.pi = method, estimatePi, Pi
    g.multi_trait_sampler = multi_trait_sampler
    return g


class ModelTerm:
    def __init__(self, trait, name):
        self.trait, self.name = trait, name
        self.kind = "intercept" if name == "intercept" else "factor"      # factor | covariate | intercept


class Model:
    """The slice of MME (types.jl:264-346) the marker path needs."""

    def __init__(self, equations, traits, terms, M, R):
        self.model_equations = equations
        self.lhsVec = traits
        self.nModels = len(traits)
        self.modelTerms = terms                # list (per trait) of ModelTerm
        self.M = M                             # list of Genotypes
        self.R = R
        self.covVec = []
        self.MCMCinfo = None
        self.output = None
        self.output_ID = False                 # outputEBV(model, IDs); False = all genotyped individuals
        self.outputSamplesVec = []             # outputMCMCsamples(model, terms...): (trait, term) pairs


def build_model(model_equations, R=False, *, df=4.0, estimate_variance=True, estimate_scale=False,
                constraint=False, genotypes=None, **unsupported):
    """build_MME.jl:42-156.  Genotype terms are found the way the reference does it -- by looking the
    term name up among the caller's variables (build_MME.jl:88-120 reflects on Main) -- or through an
    explicit `genotypes={"geno": obj}` mapping."""
    for k in unsupported:
        raise NotImplementedError(f"build_model argument '{k}' (neural-network / censored / categorical models) "
                                  "stays on the reference path")
    if not _is_false(R):                                                                      # :50-52
        Rm = np.atleast_2d(np.asarray(R, dtype=np.float64))
        ok = Rm.shape[0] == Rm.shape[1] and np.allclose(Rm, Rm.T)
        if ok:
            try:
                np.linalg.cholesky(Rm)
            except np.linalg.LinAlgError:
                ok = False
        if not ok:
            raise ValueError("The covariance matrix is not positive definite.")
    if not isinstance(model_equations, str) or model_equations.strip() == "":
        raise ValueError("Model equations are wrong.\n To find an example, type ?build_model and press enter.")   # :53-56
    if estimate_scale is not False:
        raise ValueError("estimate scale for residual variance is not supported now.")        # :57-59
    caller = inspect.currentframe().f_back
    scope = dict(caller.f_globals)
    scope.update(caller.f_locals)
    if genotypes:
        scope.update(genotypes)
    eqs = [e.strip() for e in model_equations.replace("\n", ";").split(";") if e.strip()]
    traits, terms, M = [], [], []
    for eq in eqs:
        lhs, rhs = [s.strip() for s in eq.split("=")]
        traits.append(lhs)
        tl = []
        for name in [t.strip() for t in rhs.split("+")]:
            if "*" in name:
                raise NotImplementedError("interaction terms stay on the reference path")
            obj = scope.get(name)
            if isinstance(obj, Genotypes):                                                    # :91-96
                if obj not in M:
                    obj.name = name
                    M.append(obj)
            else:
                tl.append(ModelTerm(lhs, name))
        terms.append(tl)
    if len(M) > 1:
        raise NotImplementedError("one genotype category per model on the device path (reference: 'now only work for one geno')")
    nModels = len(traits)
    if not _is_false(R) and np.atleast_2d(np.asarray(R)).shape[0] != nModels:                 # :67-69
        raise ValueError(f"The residual covariance matrix is not a {nModels} by {nModels} matrix.")
    for Mi in M:                                                                              # :98-116
        Mi.ntraits = nModels
        Mi.trait_names = traits
        if Mi.multi_trait_sampler == "II":
            if Mi.method != "BayesC":
                raise ValueError("multi_trait_sampler overrides are supported for BayesC only.")
            if nModels <= 1:
                raise ValueError("multi_trait_sampler overrides require multi-trait BayesC.")
        for V in (Mi.G, Mi.genetic_variance):
            if not _is_false(V.val) and np.atleast_2d(np.asarray(V.val)).shape[0] != nModels:
                raise ValueError(f"The genomic covariance matrix is not a {nModels} by {nModels} matrix.")
        if nModels != 1:
            Mi.G.df = Mi.G.df + nModels
            Mi.genetic_variance.df = Mi.genetic_variance.df + nModels
    if nModels == 1:                                                                          # :128-134
        Rv = False if _is_false(R) else np.float32(R)
        scale_R = False if _is_false(R) else float(R) * (df - 2) / df
        df_R = df
    else:
        Rv = False if _is_false(R) else np.asarray(R, dtype=np.float32)
        scale_R = False if _is_false(R) else np.asarray(R, dtype=np.float64) * (df - 1)
        df_R = df + nModels
    return Model(model_equations, traits, terms, M, Variance(Rv, np.float32(df_R), scale_R, estimate_variance, estimate_scale, constraint))


def outputEBV(model, IDs):
    """output.jl:60-70: individuals of interest for EBV output (default: all genotyped individuals)."""
    model.output_ID = [str(i) for i in IDs]


def outputMCMCsamples(model, *terms):
    """output.jl:76-95: also save the MCMC samples of these location-parameter terms (MCMC_samples_<trait>.<term>.txt)."""
    for trm in terms:
        for tr, tl in zip(model.lhsVec, model.modelTerms):
            if any(mt.name == trm for mt in tl) and (tr, trm) not in model.outputSamplesVec:
                model.outputSamplesVec.append((tr, trm))


def set_covariate(model, *names):
    """build_MME.jl:158-181: declare terms as continuous covariates (default is a class factor)."""
    flat = []
    for n in names:
        flat.extend(n.split())
    model.covVec.extend(flat)
    for tl in model.modelTerms:
        for t in tl:
            if t.name in flat:
                t.kind = "covariate"


def runMCMC(model, df, *, heterogeneous_residuals=False, chain_length=100, starting_value=False, burnin=0,
            output_samples_frequency=None, update_priors_frequency=0, single_step_analysis=False, pedigree=False,
            causal_structure=False, missing_phenotypes=True, RRM=False, outputEBV=True, output_heritability=True,
            prediction_equation=False, seed=False, printout_model_info=True, printout_frequency=None,
            big_memory=False, double_precision=False, fast_blocks=False, independent_blocks=False,
            memory_guard="error", memory_guard_ratio=0.80, output_folder="results",
            output_samples_for_all_parameters=False,
            # device options (the analogue of storage=:stream's opt-in knobs)
            device=0, block_size=None, gram_mode="mfma", blocks_per_launch=None, _engine=None):
    """JWAS.jl:161-511.  Returns the reference's output Dict (output.jl:108-212) as a dict of pandas
    DataFrames and writes the same text files under `output_folder`.

    fast_blocks: False -> the exact non-block chain (computed on the device in blocks with one
    within-block pass, algebraically identical: SURVEY.md section 7 step 4); True / number -> the
    reference's fast_blocks schedule (block-size repetitions, chain_length rescaled, JWAS.jl:293-316).
    independent_blocks=True (needs fast_blocks): every block starts from the same residual snapshot and all blocks are
    sampled concurrently on the device (BayesABC.jl:190-255) -- the reference's approximate parallel mode.
    `_engine` (private, not part of the reference's surface) lets the test-suite inject a sweep engine; the default and
    only shipped engine is HipEngine.
    blocks_per_launch (device option, like block_size): None = by the chain's length (mcmc.grouped_blocks_for_chain: 2 from
    3 000 iterations on, 4 from 8 000), 0 / 2 / 4 = one / two / four 1024-marker blocks per launch of the step kernel in the
    sparse steady state of a single-trait chain (grouped launches) -- the same chain up to float32
    rounding of the block right-hand sides; the group cross-Grams are set-up work (8 / 24 KB per marker).

    double_precision=True (JWAS.jl:349-366): genotypes, residual, effects and the samplers' arithmetic all Float64 -- a
    Float64 device context (jwas_hip_set_precision; csrc/f64_path.hpp): single-trait BayesA/B/C, RR-BLUP, BayesL, BayesR and
    multi-trait BayesC sampler I on dense storage; fast_blocks = 64 | 128."""
    if independent_blocks and fast_blocks is False:
        raise ValueError("independent_blocks=true requires fast_blocks != false.")             # :242-244
    for flag, name in ((single_step_analysis, "single_step_analysis"),
                       (causal_structure, "causal_structure"), (RRM, "RRM"),
                       (update_priors_frequency, "update_priors_frequency"),
                       (prediction_equation, "prediction_equation")):
        if not _is_false(flag) and flag != 0:
            raise NotImplementedError(f"runMCMC(...; {name}=...) is outside the device marker path and stays on the reference")
    if not model.M:
        raise NotImplementedError("models without a genotype term have no marker sweep: use the reference")
    if memory_guard not in ("error", "warn", "off"):
        raise ValueError("memory_guard must be :error, :warn or :off.")
    if output_samples_frequency is None:
        output_samples_frequency = chain_length // 1000 if chain_length > 1000 else 1          # :168
    if int(output_samples_frequency) <= 0:
        raise ValueError("output_samples_frequency should be an integer > 0.")                # input_data_validation.jl:14-16
    for Mi in model.M:                                                                        # :19-23
        if Mi.method not in ("BayesL", "BayesC", "BayesB", "BayesA", "BayesR", "RR-BLUP", "GBLUP"):
            raise ValueError(f"{Mi.method} is not available in JWAS. Please read the documentation.")
    if printout_frequency is None:
        printout_frequency = chain_length + 1
    # an existing output folder is never overwritten (JWAS.jl:255-262)
    myfolder, folderi = output_folder, 1
    while os.path.exists(output_folder):
        print(f"The folder {output_folder} already exists.")
        output_folder = myfolder + str(folderi)
        folderi += 1
    os.makedirs(output_folder)
    return run_chain(model, df, chain_length=int(chain_length), burnin=int(burnin),
                     output_samples_frequency=int(output_samples_frequency), seed=seed,
                     starting_value=starting_value, fast_blocks=fast_blocks,
                     independent_blocks=bool(independent_blocks), heterogeneous_residuals=bool(heterogeneous_residuals),
                     double_precision=bool(double_precision),
                     outputEBV=outputEBV, output_heritability=bool(output_heritability),
                     output_folder=output_folder, printout_frequency=printout_frequency,
                     memory_guard=memory_guard, memory_guard_ratio=memory_guard_ratio,
                     missing_phenotypes=missing_phenotypes, device=device, block_size=block_size,
                     gram_mode=gram_mode, blocks_per_launch=blocks_per_launch, engine=_engine, printout_model_info=printout_model_info,
                     output_samples_for_all_parameters=output_samples_for_all_parameters)
