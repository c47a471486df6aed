"""ctypes binding of the CPU oracle (oracle/jwas_oracle.c).  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
The shipped package (jwas.jl_amd/) never does.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libjwas_oracle.so")

ACC_F64 = 0
ACC_F32 = 1
ACC_DEVICE = 2      # x'r summed in the device's association order (set_device_order(spg) first)
GAMMA = np.array([0.0, 0.01, 0.1, 1.0], dtype=np.float64)  # JWAS.jl:12


def build(force=False):
    src = os.path.join(_HERE, "jwas_oracle.c")
    src64 = os.path.join(_HERE, "jwas_oracle_f64.c")
    hdr = os.path.join(_HERE, "jwas_oracle.h")
    stale = (not os.path.exists(_LIB_PATH)) or any(
        os.path.exists(f) and os.path.getmtime(f) > os.path.getmtime(_LIB_PATH) for f in (src, src64, hdr))
    if force or stale:
        subprocess.check_call(["make", "-C", _HERE, "-B", "libjwas_oracle.so"],
                              stdout=subprocess.DEVNULL)
    return _LIB_PATH


_lib = None

_f32p = C.POINTER(C.c_float)
_f64p = C.POINTER(C.c_double)
_i32p = C.POINTER(C.c_int32)
_i64p = C.POINTER(C.c_int64)
_u8p = C.POINTER(C.c_uint8)
_u32p = C.POINTER(C.c_uint32)


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB_PATH)
        L.orc_uniform.restype = C.c_double
        L.orc_uniform.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32]
        L.orc_normal.restype = C.c_double
        L.orc_normal.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32]
        L.orc_philox4x32_10.restype = None
        L.orc_philox4x32_10.argtypes = [_u32p, _u32p, _u32p]
        L.orc_time_bayesc_sweeps.restype = C.c_double
        L.orc_time_sweeps_team.restype = C.c_double
        L.orc_bayesr_block_nreps.argtypes = [C.c_int64, C.c_int64, C.c_int64]
        _lib = L
    return _lib


def _p(a, ty):
    return a.ctypes.data_as(ty)


def _f32(a):
    a = np.ascontiguousarray(a, dtype=np.float32)
    return a


def philox(ctr, key):
    c = np.asarray(ctr, dtype=np.uint32)
    k = np.asarray(key, dtype=np.uint32)
    out = np.zeros(4, dtype=np.uint32)
    lib().orc_philox4x32_10(_p(c, _u32p), _p(k, _u32p), _p(out, _u32p))
    return out


def uniform(seed, marker, it, rep=0, trait=0):
    return lib().orc_uniform(seed, marker, it, rep, trait)


def normal(seed, marker, it, rep=0, trait=0):
    return lib().orc_normal(seed, marker, it, rep, trait)


def _xinfo(X):
    """X: numpy array n x p, Fortran (column-major, marker-major) float32."""
    assert X.dtype == np.float32 and X.ndim == 2 and X.flags.f_contiguous
    n, p = X.shape
    ld = X.strides[1] // 4 if p > 1 else n
    return n, p, max(ld, n)


def xpx(X, acc=ACC_F64):
    n, p, ld = _xinfo(X)
    out = np.zeros(p, dtype=np.float32)
    lib().orc_xpx(_p(X, _f32p), C.c_int64(n), C.c_int64(p), C.c_int64(ld), _p(out, _f32p), C.c_int(acc))
    return out


def gram(X, j0, b, acc=ACC_F64):
    n, p, ld = _xinfo(X)
    out = np.zeros((b, b), dtype=np.float32)
    lib().orc_gram(_p(X, _f32p), C.c_int64(n), C.c_int64(ld), C.c_int64(j0), C.c_int64(b),
                   _p(out, _f32p), C.c_int(acc))
    return out


def block_starts_for(p, block_size):
    return np.arange(0, p, block_size, dtype=np.int64)


def cross_gram(X, jp, bp, j0, b, acc=ACC_F64):
    """X_prev' X_this (bp x b) with the inner products the lookahead correction uses."""
    n, p, ld = _xinfo(X)
    out = np.empty((bp, b), dtype=np.float32)
    lib().orc_cross_gram(_p(X, _f32p), C.c_int64(n), C.c_int64(ld), C.c_int64(jp), C.c_int64(bp), C.c_int64(j0), C.c_int64(b),
                         _p(out, _f32p), C.c_int(acc))
    return out


def grams_for(X, block_starts, acc=ACC_F64):
    p = X.shape[1]
    bs = list(block_starts) + [p]
    return np.concatenate([gram(X, bs[i], bs[i + 1] - bs[i], acc).ravel() for i in range(len(bs) - 1)])


def residual_minus_xalpha(X, alpha, r):
    n, p, ld = _xinfo(X)
    alpha = _f32(alpha)
    lib().orc_residual_minus_xalpha(_p(X, _f32p), C.c_int64(n), C.c_int64(p), C.c_int64(ld),
                                    _p(alpha, _f32p), _p(r, _f32p))


def _vec_or_fill(v, p, dtype):
    v = np.asarray(v, dtype=dtype)
    if v.ndim == 0:
        return np.full(p, v, dtype=dtype)
    return np.ascontiguousarray(v)


RULE_D = True        # (tests switch the rule off to compare with the literal operation order)


def set_packed_source(codes, means, centered, X):
    """The block right-hand sides in the 2-bit packed update role's own order (jwas_oracle.c dot_xr): codes (n x p, 0..3; 3 =
    missing) of the matrix X (the decoded, Fortran-ordered float32 matrix the sweeps are called with).  codes=None switches it
    off.  The arrays are borrowed: keep them alive while the mode is on."""
    global _PK_KEEP
    if codes is None:
        lib().orc_set_packed_source(None, None, C.c_int(1), None, C.c_int64(0), C.c_int64(0))
        _PK_KEEP = None
        return
    cc = np.ascontiguousarray(np.asarray(codes, dtype=np.uint8).T)          # [p][n]
    mm = np.ascontiguousarray(means, dtype=np.float32)
    n, p, ld = _xinfo(X)
    _PK_KEEP = (cc, mm, X)
    lib().orc_set_packed_source(cc.ctypes.data_as(C.c_void_p), mm.ctypes.data_as(C.c_void_p), C.c_int(1 if centered else 0),
                                X.ctypes.data_as(C.c_void_p), C.c_int64(n), C.c_int64(ld))


_PK_KEEP = None


def set_section_solve(on):
    """Rule T (the device's section_solve): dense sections of the lookahead forms' full blocks as triangular solves."""
    lib().orc_set_section_solve(C.c_int(1 if on else 0))


def set_lookahead_group(m):
    """Grouped lookahead (the device's grouped launches, group_launch): m = 2 or 4 blocks per group; 1 = the one-block lookahead."""
    lib().orc_set_lookahead_group(C.c_int(int(m)))


def section_solve_counts(reset=False):
    a, b = C.c_int64(0), C.c_int64(0)
    lib().orc_section_solve_counts(C.byref(a), C.byref(b), C.c_int(1 if reset else 0))
    return int(a.value), int(b.value)


def section_solve_exceptions():
    """Exceptions (markers taken by their literal evaluation inside a solved section) since the last reset of the counts."""
    f = lib().orc_section_solve_exceptions
    f.restype = C.c_int64
    return int(f())


def bayesabc_sweep(X, xpx_, r, alpha, beta, delta, vare, var_effects, pi, seed, it,
                   marker0=0, acc=ACC_F64, block_starts=None, grams=None, nreps=1, lookahead=False, independent=False):
    """In-place sweep.  block_starts=None -> non-block form (BayesABC.jl:60-80); lookahead=True ->
    the one-block lookahead schedule of the block form (what the HIP path runs)."""
    n, p, ld = _xinfo(X)
    ve = _vec_or_fill(var_effects, p, np.float32)
    pv = np.asarray(pi, dtype=np.float64)
    # Rule D (jwas_oracle.c abc_update; the device's definition for sweeps under a UNIFORM prior pi = 0): on exactly when the
    # sweep's pi is the scalar 0 -- a per-marker pi vector never triggers it, as on the device; RULE_D = False = literal order
    lib().orc_set_abc_rule_d(1 if (RULE_D and pv.ndim == 0 and float(pv) == 0.0) else 0)
    if pv.ndim == 1 and pv.shape[0] != p:
        # bayesabc_pi_vector (BayesABC.jl:16-22)
        raise ValueError(f"BayesABC pi vector length {pv.shape[0]} must match the number of markers ({p}).")
    pv = _vec_or_fill(pv, p, np.float64)
    for a in (r, alpha, beta, delta):
        assert a.dtype == np.float32 and a.flags.c_contiguous
    if block_starts is None:
        rc = lib().orc_bayesabc_sweep(_p(X, _f32p), C.c_int64(n), C.c_int64(p), C.c_int64(ld), _p(xpx_, _f32p),
                                      _p(r, _f32p), _p(alpha, _f32p), _p(beta, _f32p), _p(delta, _f32p),
                                      C.c_float(vare), _p(ve, _f32p), _p(pv, _f64p),
                                      C.c_uint64(seed), C.c_uint32(it), C.c_uint32(marker0), C.c_int(acc))
    else:
        bs = np.ascontiguousarray(block_starts, dtype=np.int64)
        g = _f32(grams)
        fn = lib().orc_bayesabc_indep_sweep if independent else (lib().orc_bayesabc_lookahead_sweep if lookahead else lib().orc_bayesabc_block_sweep)
        rc = fn(_p(X, _f32p), C.c_int64(n), C.c_int64(p), C.c_int64(ld), _p(xpx_, _f32p),
                                            _p(bs, _i64p), C.c_int64(len(bs)), _p(g, _f32p),
                                            _p(r, _f32p), _p(alpha, _f32p), _p(beta, _f32p), _p(delta, _f32p),
                                            C.c_float(vare), _p(ve, _f32p), _p(pv, _f64p), C.c_int(nreps),
                                            C.c_uint64(seed), C.c_uint32(it), C.c_uint32(marker0), C.c_int(acc))
    lib().orc_set_abc_rule_d(0)
    if rc != 0:
        raise ValueError(f"oracle BayesABC sweep rejected its arguments (rc={rc})")


def bayesr_sweep(X, xpx_, r, alpha, delta, vare, sigma_sq, pi, seed, it, gamma=GAMMA,
                 marker0=0, acc=ACC_F64, block_starts=None, grams=None, nreps=1, lookahead=False, independent=False):
    n, p, ld = _xinfo(X)
    pv = np.ascontiguousarray(pi, dtype=np.float64)
    is_mat = int(pv.ndim == 2)
    # bayesr_validate_priors (BayesR.jl:9-20)
    if is_mat:
        if pv.shape[0] != p:
            raise ValueError("BayesR per-marker pi must have one row per marker.")
        if pv.shape[1] != 4:
            raise ValueError("BayesR per-marker pi must have 4 columns.")
    elif pv.shape[0] != 4:
        raise ValueError(f"BayesR pi vector length {pv.shape[0]} must match the number of mixture classes (4).")
    assert delta.dtype == np.int32 and alpha.dtype == np.float32 and r.dtype == np.float32
    g4 = np.ascontiguousarray(gamma, dtype=np.float64)
    if block_starts is None:
        rc = lib().orc_bayesr_sweep(_p(X, _f32p), C.c_int64(n), C.c_int64(p), C.c_int64(ld), _p(xpx_, _f32p),
                                    _p(r, _f32p), _p(alpha, _f32p), _p(delta, _i32p),
                                    C.c_float(vare), C.c_float(sigma_sq), _p(pv, _f64p), C.c_int(is_mat),
                                    _p(g4, _f64p), C.c_uint64(seed), C.c_uint32(it), C.c_uint32(marker0), C.c_int(acc))
    else:
        bs = np.ascontiguousarray(block_starts, dtype=np.int64)
        g = _f32(grams)
        fn = lib().orc_bayesr_indep_sweep if independent else (lib().orc_bayesr_lookahead_sweep if lookahead else lib().orc_bayesr_block_sweep)
        rc = fn(_p(X, _f32p), C.c_int64(n), C.c_int64(p), C.c_int64(ld), _p(xpx_, _f32p),
                                          _p(bs, _i64p), C.c_int64(len(bs)), _p(g, _f32p),
                                          _p(r, _f32p), _p(alpha, _f32p), _p(delta, _i32p),
                                          C.c_float(vare), C.c_float(sigma_sq), _p(pv, _f64p), C.c_int(is_mat),
                                          _p(g4, _f64p), C.c_int(nreps),
                                          C.c_uint64(seed), C.c_uint32(it), C.c_uint32(marker0), C.c_int(acc))
    if rc == -2:
        raise ValueError("BayesR sigmaSq must be positive.")
    if rc != 0:
        raise ValueError("BayesR pi entries must be nonnegative and sum to 1.")


def bayesr_block_nreps(it, burnin, block_size):
    v = lib().orc_bayesr_block_nreps(it, burnin, block_size)
    if v < 0:
        raise ValueError("BayesR block_size must be at least 1.")
    return v


def bayesr_sigma_suffstats(alpha, delta, gamma=GAMMA):
    a = _f32(alpha)
    d = np.ascontiguousarray(delta, dtype=np.int32)
    g = np.ascontiguousarray(gamma, dtype=np.float64)
    ssq = C.c_double(0.0)
    nnz = C.c_int64(0)
    lib().orc_bayesr_sigma_suffstats(_p(a, _f32p), _p(d, _i32p), C.c_int64(len(a)), _p(g, _f64p),
                                     C.byref(ssq), C.byref(nnz))
    return ssq.value, nnz.value


MT_SAMPLER_I, MT_SAMPLER_II, MT_MEGA = 1, 2, 3


def mt_sweep(kind, X, xpx_, r, alpha, beta, delta, vare, var_effect, log_prior, seed, it,
             marker0=0, acc=ACC_F64, block_starts=None, grams=None, nreps=1, lookahead=False, independent=False):
    """Multi-trait sweep, kind = MT_SAMPLER_I / MT_SAMPLER_II / MT_MEGA.
    r: t x ld_r float32 C-contiguous; alpha/beta/delta: t x p float32 C-contiguous.
    log_prior: 2^t (global) or p x 2^t (marker-specific) float64; for MT_MEGA the t per-trait pi values."""
    n, p, ld = _xinfo(X)
    t = r.shape[0]
    ld_r = r.shape[1]
    ve = np.ascontiguousarray(vare, dtype=np.float32)
    vg = np.ascontiguousarray(var_effect, dtype=np.float32)
    lp = np.ascontiguousarray(log_prior, dtype=np.float64)
    is_mat = int(lp.ndim == 2)
    for a in (r, alpha, beta, delta):
        assert a.dtype == np.float32 and a.flags.c_contiguous
    if block_starts is None:
        rc = lib().orc_mt_sweep(C.c_int(kind), _p(X, _f32p), C.c_int64(n), C.c_int64(p), C.c_int64(ld), _p(xpx_, _f32p),
                                C.c_int(t), _p(r, _f32p), C.c_int64(ld_r),
                                _p(alpha, _f32p), _p(beta, _f32p), _p(delta, _f32p),
                                _p(ve, _f32p), _p(vg, _f32p), _p(lp, _f64p), C.c_int(is_mat),
                                C.c_uint64(seed), C.c_uint32(it), C.c_uint32(marker0), C.c_int(acc))
    else:
        bs = np.ascontiguousarray(block_starts, dtype=np.int64)
        g = _f32(grams)
        fn = lib().orc_mt_indep_sweep if independent else (lib().orc_mt_lookahead_sweep if lookahead else lib().orc_mt_block_sweep)
        rc = fn(C.c_int(kind), _p(X, _f32p), C.c_int64(n), C.c_int64(p), C.c_int64(ld), _p(xpx_, _f32p),
                _p(bs, _i64p), C.c_int64(len(bs)), _p(g, _f32p),
                C.c_int(t), _p(r, _f32p), C.c_int64(ld_r),
                _p(alpha, _f32p), _p(beta, _f32p), _p(delta, _f32p),
                _p(ve, _f32p), _p(vg, _f32p), _p(lp, _f64p), C.c_int(is_mat),
                C.c_int(nreps),
                C.c_uint64(seed), C.c_uint32(it), C.c_uint32(marker0), C.c_int(acc))
    if rc != 0:
        raise ValueError(f"oracle multi-trait sweep (kind {kind}) rejected its arguments (rc={rc})")


def set_device_order(spg):
    """Slices per row group of the device context whose summation order ACC_DEVICE mirrors (HipEngine.update_geometry())."""
    lib().orc_set_device_order(C.c_int(int(spg)))


def set_var_effect_matrix(mat):
    """Per-marker effect covariances (p x t x t float32) for the multi-trait sweeps (multi-trait BayesA/B), or None."""
    global _VEM
    if mat is None:
        _VEM = None
        lib().orc_set_var_effect_matrix(None)
    else:
        _VEM = np.ascontiguousarray(mat, dtype=np.float32)
        lib().orc_set_var_effect_matrix(_p(_VEM, _f32p))


_VEM = None


def sample_marker_covariances(beta, df, scale, seed, it, marker0=0, diagonal=False):
    """One InverseWishart(df, scale + b_j b_j') draw per marker (variance_components.jl:181-186; df = the reference's
    df + 1): beta t x p float32, scale t x t -> p x t x t float32 (orc_sample_marker_covariances).  diagonal=True
    (constraint = true, :112-117): G_kk = (scale_kk + b_jk^2) / chi2(df) and zero off-diagonals."""
    b = np.ascontiguousarray(beta, dtype=np.float32)
    t, p = b.shape
    sc = np.ascontiguousarray(scale, dtype=np.float64).reshape(t, t)
    out = np.empty((p, t, t), dtype=np.float32)
    L = lib()
    fn = L.orc_sample_marker_variances_diag if diagonal else L.orc_sample_marker_covariances
    fn.restype = None
    fn.argtypes = [C.c_int, C.c_int64, _f32p, C.c_double, _f64p, C.c_uint64, C.c_uint32, C.c_uint32, _f32p]
    fn(t, p, _p(b, _f32p), float(df), _p(sc, _f64p), int(seed), int(it), int(marker0), _p(out, _f32p))
    return out


def mtbayesc_I_sweep(X, xpx_, r, alpha, beta, delta, vare, var_effect, log_prior, seed, it, **kw):
    mt_sweep(MT_SAMPLER_I, X, xpx_, r, alpha, beta, delta, vare, var_effect, log_prior, seed, it, **kw)


def accumulate(alpha, delta, k, mean_alpha, mean_alpha2, mean_delta):
    a = _f32(alpha)
    is_class = int(delta.dtype == np.int32)
    d = np.ascontiguousarray(delta)
    lib().orc_accumulate(_p(a, _f32p), d.ctypes.data_as(C.c_void_p), C.c_int(is_class), C.c_int64(len(a)),
                         C.c_double(k), _p(mean_alpha, _f32p), _p(mean_alpha2, _f32p), _p(mean_delta, _f32p))


_weights_keepalive = None


def set_weights(rinv):
    """Residual weights R^-1 for every inner product of the oracle (None = unit).  The array is kept alive here."""
    global _weights_keepalive
    if rinv is None:
        _weights_keepalive = None
        lib().orc_set_weights(None)
        return
    _weights_keepalive = np.ascontiguousarray(rinv, dtype=np.float32)
    lib().orc_set_weights(_p(_weights_keepalive, _f32p))


def decode_marker_2bit(payload, n, j, mean, centered=True):
    pl = np.ascontiguousarray(payload, dtype=np.uint8)
    out = np.zeros(n, dtype=np.float32)
    lib().orc_decode_marker_2bit(_p(pl, _u8p), C.c_int64(n), C.c_int64(j), C.c_float(mean),
                                 C.c_int(int(centered)), _p(out, _f32p))
    return out


def time_bayesc_sweeps(X, xpx_, r, alpha, beta, delta, vare, var_effect, pi, seed, sweeps, nthreads=1):
    n, p, ld = _xinfo(X)
    return lib().orc_time_bayesc_sweeps(_p(X, _f32p), C.c_int64(n), C.c_int64(p), C.c_int64(ld), _p(xpx_, _f32p),
                                        _p(r, _f32p), _p(alpha, _f32p), _p(beta, _f32p), _p(delta, _f32p),
                                        C.c_float(vare), C.c_float(var_effect), C.c_double(pi),
                                        C.c_uint64(seed), C.c_int(sweeps), C.c_int(nthreads))


def time_sweeps_team(kind, X, xpx_, r, alpha, beta, delta, vare, var_effect, prior, seed, sweeps, nthreads=1, gamma=GAMMA,
                     max_seconds=0.0):
    """kind 0 = BayesC, 1 = BayesR, 2 = multi-trait sampler I; state arrays t x p (1-D for one trait), r t x ld_r.
    Runs up to `sweeps` non-block sweeps with a persistent team of `nthreads` threads (stops after max_seconds if > 0).
    Returns (elapsed seconds, marker updates completed)."""
    n, p, ld = _xinfo(X)
    r2 = r.reshape(1, -1) if r.ndim == 1 else r
    t = r2.shape[0]
    ve = np.ascontiguousarray(vare, dtype=np.float32).reshape(-1)
    vg = np.ascontiguousarray(var_effect, dtype=np.float32).reshape(-1)
    pr = np.ascontiguousarray(prior, dtype=np.float64).reshape(-1)
    g4 = np.ascontiguousarray(gamma, dtype=np.float64)
    assert r2.dtype == np.float32 and alpha.dtype == np.float32 and r2.flags.c_contiguous
    bt = beta if beta is not None else alpha
    done = C.c_int64(0)
    el = lib().orc_time_sweeps_team(C.c_int(kind), _p(X, _f32p), C.c_int64(n), C.c_int64(p), C.c_int64(ld), _p(xpx_, _f32p),
                                    C.c_int(t), _p(r2, _f32p), C.c_int64(r2.shape[1]), _p(alpha, _f32p), _p(bt, _f32p),
                                    delta.ctypes.data_as(C.c_void_p), _p(ve, _f32p), _p(vg, _f32p), _p(pr, _f64p), _p(g4, _f64p),
                                    C.c_uint64(seed), C.c_int(sweeps), C.c_int(nthreads), C.c_double(max_seconds), C.byref(done))
    if el < 0:
        raise ValueError("oracle timing helper rejected its arguments")
    return el, done.value


# ---- Float64 mode (runMCMC(double_precision=true)): oracle/jwas_oracle_f64.c ------------------------------------------
def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def xpx64(X):
    X = np.asfortranarray(X, dtype=np.float64)
    n, p = X.shape
    out = np.empty(p)
    L = lib()
    L.orc64_xpx.restype = None
    L.orc64_xpx.argtypes = [_f64p, C.c_int64, C.c_int64, C.c_int64, _f64p]
    L.orc64_xpx(_p(X, _f64p), n, p, n, _p(out, _f64p))
    return out


def bayesabc_sweep64(X, xpx_, r, alpha, beta, delta, vare, var_effects, pi, seed, it, marker0=0, block_size=0, nreps=1):
    """BayesABC! with T = Float64, in place; block_size > 0: BayesABC_block! with `nreps` repetitions."""
    n, p = X.shape
    ve = np.broadcast_to(np.asarray(var_effects, dtype=np.float64), (p,)).copy()
    pv = np.broadcast_to(np.asarray(pi, dtype=np.float64), (p,)).copy()
    L = lib()
    common = [_p(r, _f64p), _p(alpha, _f64p), _p(beta, _f64p), _p(delta, _f64p), C.c_double(float(vare)), _p(ve, _f64p), _p(pv, _f64p),
              C.c_uint64(int(seed)), C.c_uint32(int(it)), C.c_uint32(int(marker0))]
    if block_size:
        rc = L.orc64_bayesabc_block_sweep(_p(X, _f64p), C.c_int64(n), C.c_int64(p), C.c_int64(n), _p(xpx_, _f64p), C.c_int64(int(block_size)),
                                          C.c_int(int(nreps)), *common)
    else:
        rc = L.orc64_bayesabc_sweep(_p(X, _f64p), C.c_int64(n), C.c_int64(p), C.c_int64(n), _p(xpx_, _f64p), *common)
    if rc:
        raise ValueError(f"orc64_bayesabc_sweep: {rc}")


def xpx64_w(X, w=None):
    """x'R^-1 x in Float64 (w = None: unit weights)."""
    X = np.asfortranarray(X, dtype=np.float64)
    n, p = X.shape
    out = np.empty(p)
    L = lib()
    L.orc64_xpx_w.restype = None
    wv = None if w is None else _f64(w)
    L.orc64_xpx_w(_p(X, _f64p), C.c_int64(n), C.c_int64(p), C.c_int64(n), None if wv is None else _p(wv, _f64p), _p(out, _f64p))
    return out


def bayesabc_block_sweep64_ex(X, xpx_, r, alpha, beta, delta, vare, var_effects, pi, seed, it, starts, nreps=1, independent=False,
                              w=None, marker0=0):
    """BayesABC_block! / BayesABC_block_independent! with T = Float64 on the partition `starts` (0-based block starts; the
    end p is appended here), residual weights w, in place."""
    n, p = X.shape
    ve = np.broadcast_to(np.asarray(var_effects, dtype=np.float64), (p,)).copy()
    pv = np.broadcast_to(np.asarray(pi, dtype=np.float64), (p,)).copy()
    st = np.ascontiguousarray(list(starts) + [p], dtype=np.int64)
    wv = None if w is None else _f64(w)
    rc = lib().orc64_bayesabc_block_sweep_ex(_p(X, _f64p), C.c_int64(n), C.c_int64(p), C.c_int64(n), _p(xpx_, _f64p),
                                             None if wv is None else _p(wv, _f64p), st.ctypes.data_as(C.POINTER(C.c_int64)), C.c_int64(len(st) - 1),
                                             C.c_int(int(nreps)), C.c_int(1 if independent else 0),
                                             _p(r, _f64p), _p(alpha, _f64p), _p(beta, _f64p), _p(delta, _f64p), C.c_double(float(vare)),
                                             _p(ve, _f64p), _p(pv, _f64p), C.c_uint64(int(seed)), C.c_uint32(int(it)), C.c_uint32(int(marker0)))
    if rc:
        raise ValueError(f"orc64_bayesabc_block_sweep_ex: {rc}")


def bayesr_sweep64(X, xpx_, r, alpha, delta, vare, sigma_sq, pi, seed, it, gamma=GAMMA, marker0=0):
    n, p = X.shape
    pm = _f64(pi)
    g = _f64(gamma)
    rc = lib().orc64_bayesr_sweep(_p(X, _f64p), C.c_int64(n), C.c_int64(p), C.c_int64(n), _p(xpx_, _f64p), _p(r, _f64p), _p(alpha, _f64p),
                                  _p(delta, _i32p), C.c_double(float(vare)), C.c_double(float(sigma_sq)), _p(pm, _f64p), C.c_int(int(pm.ndim == 2)),
                                  _p(g, _f64p), C.c_uint64(int(seed)), C.c_uint32(int(it)), C.c_uint32(int(marker0)))
    if rc:
        raise ValueError(f"orc64_bayesr_sweep: {rc}")


def mt1_sweep64(X, xpx_, r, alpha, beta, delta, vare, var_effect, log_prior, seed, it, marker0=0):
    """_MTBayesABC_samplerI! with T = Float64; r: t x n, alpha / beta / delta: t x p (all C-contiguous float64, in place)."""
    n, p = X.shape
    t = r.shape[0]
    rc = lib().orc64_mt1_sweep(C.c_int(t), _p(X, _f64p), C.c_int64(n), C.c_int64(p), C.c_int64(n), _p(xpx_, _f64p), _p(r, _f64p), C.c_int64(n),
                               _p(alpha, _f64p), _p(beta, _f64p), _p(delta, _f64p), _p(_f64(vare), _f64p), _p(_f64(var_effect), _f64p),
                               _p(_f64(log_prior), _f64p), C.c_uint64(int(seed)), C.c_uint32(int(it)), C.c_uint32(int(marker0)))
    if rc:
        raise ValueError(f"orc64_mt1_sweep: {rc}")
