"""HipEngine: numpy-facing wrapper over the C ABI -- the device side of the marker sweep.

The host MCMC loop (jwas.jl_amd/mcmc.py) drives a *sweep engine* through this small protocol:

    load_dense(X) / alloc_dense(n,p) + synth(...)      genotype storage      (Genotypes.genotypes)
    setup_blocks(block_size, gram_mode)                x'x + block Grams     (GibbsMats)
    xpx() / gram(i) / set_gram(i, G)
    init_state(method, ntraits); set_state/get_state   alpha, beta, delta
    set_residual/get_residual(trait)                   ycorr
    sub_xalpha(trait); mul_alpha(trait)
    sweep(**params) -> dict of reductions              BayesABC!/BayesR!/MTBayesABC!
    accumulate(k); posterior(trait)                    running posterior means

HipEngine is the only engine the package ships; it raises if libjwas_hip.so or the GPU is missing.
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import JwasHipError, SweepParams, SweepStats

METHOD_CODES = {"BayesC": _lib.BAYESC, "BayesB": _lib.BAYESB, "BayesA": _lib.BAYESB,
                "BayesR": _lib.BAYESR, "MTBayesC": _lib.MTBAYESC1, "MTBayesC_II": _lib.MTBAYESC2,
                "MegaBayesC": _lib.MEGABAYESC, "MTBayesB": _lib.MTBAYESB1, "MTBayesB_II": _lib.MTBAYESB2, "MegaBayesB": _lib.MEGABAYESB}
BAYESR_GAMMA = np.array([0.0, 0.01, 0.1, 1.0], dtype=np.float64)   # JWAS.jl:12


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


class SectionSolvePolicy:
    """When the host sets jwas_sweep_params.section_solve (Rule T) along a chain.  A multi-trait sampler-I chain that starts from
    the reference's default prior (every marker in the model) drifts, with Pi estimated, towards sparser joint states; once
    most 64-marker sections hold more markers outside the model than the rule takes as exceptions, the per-sweep section
    inverses and the lazily fetched Gram tiles are pure overhead.  The rule is therefore switched off for `probe` sweeps when
    the last sweep solved fewer than half of its sections, and tried again afterwards (speed only: on or off, a sweep samples
    the same conditionals)."""

    def __init__(self, enabled, nsections, probe=50, probe_max=800):
        self.enabled, self.nsections, self.probe, self.probe_max = bool(enabled), int(nsections), int(probe), int(probe_max)
        self.probe0 = self.probe
        self.off_until = 0

    def use(self, it):
        return self.enabled and it >= self.off_until

    def observe(self, it, engine, ran=True):
        """After sweep `it`.  ran: the sweep really passed section_solve (a sweep on another block size did not, and its
        counter 16 means something else -- reading it would switch the rule off without evidence)."""
        if not ran or not self.enabled or it < self.off_until or not hasattr(engine, "last_sweep_counters"):
            return
        if engine.last_sweep_counters()[16] * 2 < self.nsections:
            # (a probe sweep that fails costs more than a sweep without the rule: back off -- 50, 100, 200 ... sweeps between tries)
            self.off_until = it + 1 + self.probe
            self.probe = min(2 * self.probe, self.probe_max)
        else:
            self.probe = self.probe0


class HipEngine:
    def __init__(self, device=0, precision=32):
        """precision: 32 (the reference's default Float32 path) or 64 (runMCMC(double_precision=true): a Float64 context --
        genotypes, residual, effects and the samplers' arithmetic all in double; jwas_hip_set_precision)."""
        if precision not in (32, 64):
            raise ValueError("precision must be 32 or 64")
        self._L = _lib.load()
        h = C.c_void_p()
        rc = self._L.jwas_hip_create(int(device), C.byref(h))
        if rc != 0:
            raise JwasHipError(rc, self._L.jwas_hip_last_error(None).decode())
        self._h = h
        self.precision = int(precision)
        self.dtype = np.float64 if precision == 64 else np.float32
        if precision == 64:
            rc = self._L.jwas_hip_set_precision(h, 64)
            if rc != 0:
                raise JwasHipError(rc, self._L.jwas_hip_last_error(h).decode())
        self.device = int(device)
        self.n = self.p = 0
        self.method = None
        self.ntraits = 0
        self.block_size = 0
        self._keep = []   # host arrays referenced by the last sweep call
        self._weighted = False          # non-unit residual weights on the device (set_weights)
        self._explicit_starts = None    # explicit block partition resident (setup_blocks_explicit)

    # -- plumbing --------------------------------------------------------------------------------
    def _chk(self, rc):
        if rc != 0:
            raise JwasHipError(rc, self._L.jwas_hip_last_error(self._h).decode())

    def close(self):
        if getattr(self, "_h", None):
            self._L.jwas_hip_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_stream(self, stream_handle):
        self._chk(self._L.jwas_hip_set_stream(self._h, C.c_void_p(int(stream_handle))))

    def device_info(self):
        ncu, tot, free = C.c_int(), C.c_int64(), C.c_int64()
        self._chk(self._L.jwas_hip_device_info(self._h, C.byref(ncu), C.byref(tot), C.byref(free)))
        return {"n_cu": ncu.value, "hbm_total": tot.value, "hbm_free": free.value}

    @staticmethod
    def estimate_bytes(n, p, ntraits=1, block_size=256, storage="dense"):
        kind = _lib.STORAGE_PACKED2BIT if storage in ("stream", "packed2bit") else _lib.STORAGE_DENSE_F32
        return _lib.load().jwas_hip_estimate_bytes_storage(int(n), int(p), int(ntraits), int(block_size), kind)

    # -- storage ---------------------------------------------------------------------------------
    def load_dense(self, X):
        """X: n x p float32.  Fortran order is uploaded as is (marker-major, zero re-layout)."""
        X = np.asarray(X)
        if X.dtype != self.dtype:
            raise TypeError(f"this engine stores {np.dtype(self.dtype).name} genotypes (double_precision={'true' if self.precision == 64 else 'false'})")
        if X.ndim != 2:
            raise ValueError("genotype matrix must be 2-D")
        if not X.flags.f_contiguous:
            X = np.asfortranarray(X)
        n, p = X.shape
        self._chk((self._L.jwas_hip_load_dense_f64 if self.precision == 64 else self._L.jwas_hip_load_dense_f32)(self._h, _ptr(X), n, p, n))
        self.n, self.p = n, p
        self.method, self.block_size = None, 0

    def alloc_dense(self, n, p):
        self._chk(self._L.jwas_hip_alloc_dense_f32(self._h, int(n), int(p)))
        self.n, self.p = int(n), int(p)
        self.method, self.block_size = None, 0

    # 2-bit packed storage (the reference's Packed2BitBackend kept packed in HBM) ----------------
    def load_packed2bit(self, payload, n, means, centered=True):
        """payload: p x stride uint8 (marker-major rows, stride >= cld(n,4)); means: p float32."""
        payload = np.ascontiguousarray(payload, dtype=np.uint8)
        if payload.ndim != 2:
            raise ValueError("payload must be a p x stride_bytes uint8 array")
        means = np.ascontiguousarray(means, dtype=np.float32)
        p, stride = payload.shape
        if means.shape != (p,):
            raise ValueError("one marker mean per packed marker is required")
        self._chk(self._L.jwas_hip_load_packed2bit(self._h, _ptr(payload), int(n), p, stride, _ptr(means), int(bool(centered))))
        self.n, self.p = int(n), p
        self.method, self.block_size = None, 0

    def load_jgb2(self, path):
        """load_streaming_backend (streaming_genotypes.jl:884-971): prefix, or <prefix>.meta / <prefix>.jgb2."""
        self._chk(self._L.jwas_hip_load_jgb2(self._h, str(path).encode()))
        info = self.storage_info()
        self.n, self.p = info["n"], info["p"]
        self.method, self.block_size = None, 0

    def alloc_packed(self, n, p, centered=True):
        self._chk(self._L.jwas_hip_alloc_packed2bit(self._h, int(n), int(p), int(bool(centered))))
        self.n, self.p = int(n), int(p)
        self.method, self.block_size = None, 0

    def storage_info(self):
        k, n, p, b = C.c_int32(), C.c_int64(), C.c_int64(), C.c_int64()
        self._chk(self._L.jwas_hip_storage_info(self._h, C.byref(k), C.byref(n), C.byref(p), C.byref(b)))
        return {"kind": "packed2bit" if k.value == _lib.STORAGE_PACKED2BIT else "dense_f32", "n": n.value, "p": p.value,
                "bytes": b.value}

    def set_xpx(self, xpx):
        x = np.ascontiguousarray(xpx, dtype=np.float32)
        if x.shape != (self.p,):
            raise ValueError(f"x'x must have length {self.p}")
        self._chk(self._L.jwas_hip_set_xpx(self._h, _ptr(x)))

    def set_weights(self, rinv):
        """Residual weights R^-1 = 1 ./ weights (build_MME.jl:305-310); None = unit.  Call before setup_blocks."""
        if rinv is None:
            self._chk(self._L.jwas_hip_set_weights(self._h, None))
        else:
            w = np.ascontiguousarray(rinv, dtype=self.dtype)             # Float64 contexts keep the weights' Float64 values (build_MME.jl:310)
            if w.shape != (self.n,):
                raise ValueError(f"one weight per individual is required ({self.n}), got {w.shape}")
            self._chk((self._L.jwas_hip_set_weights_f64 if self.precision == 64 else self._L.jwas_hip_set_weights)(self._h, _ptr(w)))
        self._weighted = rinv is not None
        self.block_size = 0

    def synth(self, seed, kind=0, center=True, marker_offset=0):
        self._chk(self._L.jwas_hip_synth_genotypes(self._h, int(seed), int(kind), int(bool(center)), int(marker_offset)))

    def synth_single_step(self, seed, n_genotyped, center=True, marker_offset=0):
        """Config-5 shaped input: rows < n_genotyped are 0/1/2 genotypes, the rest real-valued 'imputed' rows."""
        self._chk(self._L.jwas_hip_synth_single_step(self._h, int(seed), int(n_genotyped), int(bool(center)), int(marker_offset)))

    def layout(self):
        n, p, ld, ptr = C.c_int64(), C.c_int64(), C.c_int64(), C.c_void_p()
        self._chk(self._L.jwas_hip_dense_layout(self._h, C.byref(n), C.byref(p), C.byref(ld), C.byref(ptr)))
        return {"n": n.value, "p": p.value, "ld": ld.value, "ptr": ptr.value}

    def get_columns(self, j0, count):
        out = np.empty((self.n, int(count)), dtype=np.float32, order="F")
        self._chk(self._L.jwas_hip_get_columns(self._h, int(j0), int(count), _ptr(out)))
        return out

    def set_columns(self, j0, cols):
        """Overwrite columns [j0, j0 + cols.shape[1]) of an allocated dense matrix (n x count float32, any order)."""
        cols = np.asfortranarray(cols, dtype=np.float32)
        if cols.ndim != 2 or cols.shape[0] != self.n:
            raise ValueError(f"column chunk must be {self.n} x count")
        self._chk(self._L.jwas_hip_set_columns(self._h, int(j0), int(cols.shape[1]), _ptr(cols), self.n))
        self.block_size = 0

    # -- precompute ------------------------------------------------------------------------------
    def setup_blocks(self, block_size=256, gram_mode="mfma"):
        mode = {"f64": _lib.GRAM_F64, "mfma": _lib.GRAM_MFMA}[gram_mode]
        self._chk(self._L.jwas_hip_setup_blocks(self._h, int(block_size), mode))
        self.block_size = int(block_size)
        self._resident = [int(block_size)]
        self._groups = {}
        self._explicit_starts = None

    def setup_blocks_explicit(self, starts, gram_mode="mfma"):
        """Explicit, possibly non-uniform block partition (fast_blocks = a vector of block starts, JWAS.jl:298-304):
        `starts` = 0-based first marker of every block, starts[0] = 0, strictly increasing, blocks of <= 1024 markers."""
        mode = {"f64": _lib.GRAM_F64, "mfma": _lib.GRAM_MFMA}[gram_mode]
        st = np.ascontiguousarray(starts, dtype=np.int64)
        self._chk(self._L.jwas_hip_setup_blocks_explicit(self._h, st.ctypes.data_as(C.POINTER(C.c_int64)), len(st), mode))
        nb, bs = C.c_int64(0), C.c_int32(0)
        self._chk(self._L.jwas_hip_num_blocks(self._h, C.byref(nb), C.byref(bs)))
        self.block_size = int(bs.value)
        self._resident = [self.block_size]
        self._groups = {}
        self._explicit_starts = st.copy()

    def add_block_size(self, block_size, gram_mode="mfma"):
        """Make a second block size resident (see jwas_hip_add_block_size); select_block_size switches between sweeps."""
        mode = {"f64": _lib.GRAM_F64, "mfma": _lib.GRAM_MFMA}[gram_mode]
        self._chk(self._L.jwas_hip_add_block_size(self._h, int(block_size), mode))
        self._resident.append(int(block_size))

    def setup_groups(self, blocks_per_launch, gram_mode="mfma"):
        """Grouped launches for the selected block size (jwas_hip_setup_groups): 2 or 4 consecutive blocks per launch of the step
        kernel, used by the sweeps that pass group_launch=True; 0 frees the buffers."""
        mode = {"f64": _lib.GRAM_F64, "mfma": _lib.GRAM_MFMA}[gram_mode]
        self._chk(self._L.jwas_hip_setup_groups(self._h, int(blocks_per_launch), mode))
        self._groups = dict(getattr(self, "_groups", {}))
        self._groups[self.block_size] = int(blocks_per_launch)

    def blocks_per_launch(self, block_size=None):
        """Blocks per grouped launch set up for a block size (0: none)."""
        return getattr(self, "_groups", {}).get(self.block_size if block_size is None else int(block_size), 0)

    def resident_block_sizes(self):
        """Block sizes whose Grams are resident (setup_blocks / add_block_size)."""
        return list(getattr(self, "_resident", [])) if self.block_size else []

    def select_block_size(self, block_size):
        self._chk(self._L.jwas_hip_select_block_size(self._h, int(block_size)))
        self.block_size = int(block_size)

    @property
    def nblocks(self):
        nb, bs = C.c_int64(), C.c_int32()
        self._chk(self._L.jwas_hip_num_blocks(self._h, C.byref(nb), C.byref(bs)))
        return nb.value

    def block_starts(self):
        if getattr(self, "_explicit_starts", None) is not None:
            return self._explicit_starts.copy()
        return np.arange(0, self.p, self.block_size, dtype=np.int64)

    def _bsize(self, i):
        if getattr(self, "_explicit_starts", None) is not None:
            st = self._explicit_starts
            return int((st[i + 1] if i + 1 < len(st) else self.p) - st[i])
        j0 = i * self.block_size
        return min(self.block_size, self.p - j0)

    def xpx(self):
        out = np.empty(self.p, dtype=self.dtype)
        self._chk((self._L.jwas_hip_get_xpx_f64 if self.precision == 64 else self._L.jwas_hip_get_xpx)(self._h, _ptr(out)))
        return out

    def gram(self, i):
        b = self._bsize(i)
        out = np.empty((b, b), dtype=np.float32)
        self._chk(self._L.jwas_hip_get_gram(self._h, int(i), _ptr(out)))
        return out

    def set_gram(self, i, G):
        b = self._bsize(i)
        G = np.ascontiguousarray(G, dtype=np.float32)
        if G.shape != (b, b):
            raise ValueError(f"Gram block {i} must be {b} x {b}")
        self._chk(self._L.jwas_hip_set_gram(self._h, int(i), _ptr(G)))

    def set_cross_gram(self, i, C_):
        """Overwrite X_{i-1}' X_i (rows = markers of block i-1), i >= 1."""
        Cc = np.ascontiguousarray(C_, dtype=np.float32)
        if Cc.shape != (self._bsize(i - 1), self._bsize(i)):
            raise ValueError(f"cross-Gram of block {i} must be {self._bsize(i - 1)} x {self._bsize(i)}")
        self._chk(self._L.jwas_hip_set_cross_gram(self._h, int(i), _ptr(Cc)))

    def update_geometry(self):
        """(slices per row group, row groups, column groups) of the streaming role for this matrix."""
        a, b, c_ = C.c_int32(), C.c_int32(), C.c_int32()
        self._chk(self._L.jwas_hip_update_geometry(self._h, C.byref(a), C.byref(b), C.byref(c_)))
        return a.value, b.value, c_.value

    def set_grams_packed(self, grams):
        """grams: concatenated row-major b_i x b_i blocks (the oracle's packing)."""
        off = 0
        g = np.ascontiguousarray(grams, dtype=np.float32)
        for i in range(self.nblocks):
            b = self._bsize(i)
            self.set_gram(i, g[off:off + b * b].reshape(b, b))
            off += b * b

    # -- state -----------------------------------------------------------------------------------
    def init_state(self, method, ntraits=1):
        code = METHOD_CODES[method] if isinstance(method, str) else int(method)
        self._chk(self._L.jwas_hip_init_state(self._h, code, int(ntraits)))
        self.method, self.ntraits = code, int(ntraits)

    def _delta_dtype(self):
        return np.int32 if self.method == _lib.BAYESR else self.dtype

    def _f(self, name):
        """The entry point `name`, or its _f64 namesake in a Float64 context."""
        return getattr(self._L, name + "_f64" if self.precision == 64 else name)

    def set_state(self, trait=0, alpha=None, beta=None, delta=None):
        a = None if alpha is None else np.ascontiguousarray(alpha, dtype=self.dtype)
        b = None if beta is None else np.ascontiguousarray(beta, dtype=self.dtype)
        d = None if delta is None else np.ascontiguousarray(delta, dtype=self._delta_dtype())
        for v in (a, b, d):
            if v is not None and v.shape != (self.p,):
                raise ValueError(f"state vectors must have length {self.p}")
        self._chk(self._f("jwas_hip_set_state")(self._h, int(trait), _ptr(a), _ptr(b), _ptr(d)))

    def get_state(self, trait=0):
        a = np.empty(self.p, dtype=self.dtype)
        b = np.empty(self.p, dtype=self.dtype)
        d = np.empty(self.p, dtype=self._delta_dtype())
        self._chk(self._f("jwas_hip_get_state")(self._h, int(trait), _ptr(a), _ptr(b), _ptr(d)))
        return a, b, d

    def set_residual(self, r, trait=0):
        r = np.ascontiguousarray(r, dtype=self.dtype)
        if r.shape != (self.n,):
            raise ValueError(f"residual must have length {self.n}")
        self._chk(self._f("jwas_hip_set_residual")(self._h, int(trait), _ptr(r)))

    def get_residual(self, trait=0):
        r = np.empty(self.n, dtype=self.dtype)
        self._chk(self._f("jwas_hip_get_residual")(self._h, int(trait), _ptr(r)))
        return r

    def residual_dev(self):
        ptr, ld = C.c_void_p(), C.c_int64()
        self._chk(self._L.jwas_hip_residual_dev(self._h, C.byref(ptr), C.byref(ld)))
        return ptr.value, ld.value

    def residual_to_dev(self, dst_ptr, trait=0):
        self._chk(self._L.jwas_hip_residual_to_dev(self._h, int(trait), C.c_void_p(int(dst_ptr))))

    def residual_from_dev(self, src_ptr, trait=0):
        self._chk(self._L.jwas_hip_residual_from_dev(self._h, int(trait), C.c_void_p(int(src_ptr))))

    def residual_add_scalar(self, shift, trait=0):
        """r_trait += shift on the device (the intercept's residual correction; no host copy of the residual)."""
        self._chk(self._L.jwas_hip_residual_add_scalar(self._h, int(trait), float(shift)))

    def last_sweep_counters(self):
        """The sampler's diagnostics counters of the last sweep (jwas_hip_last_sweep_counters)."""
        out = (C.c_uint64 * 32)()
        self._chk(self._L.jwas_hip_last_sweep_counters(self._h, out, 32))
        return [int(v) for v in out]

    def set_kernel_timing(self, stride):
        self._chk(self._L.jwas_hip_set_kernel_timing(self._h, int(stride)))

    def sub_xalpha(self, trait=0):
        self._chk(self._L.jwas_hip_residual_sub_xalpha(self._h, int(trait)))

    def mul_alpha(self, trait=0):
        out = np.empty(self.n, dtype=self.dtype)
        self._chk(self._f("jwas_hip_mul_alpha")(self._h, int(trait), _ptr(out)))
        return out

    def alpha_sparse(self, trait=0):
        """(idx int32, val float32): the nonzero effects of a trait in marker order, compacted on the device."""
        cap = getattr(self, "_sparse_cap", 4096)
        while True:
            idx = np.empty(cap, dtype=np.int32)
            val = np.empty(cap, dtype=np.float32)
            nnz = C.c_int64(0)
            rc = self._L.jwas_hip_get_alpha_sparse(self._h, int(trait), cap, _ptr(idx), _ptr(val), C.byref(nnz))
            if rc == 0:
                return idx[:nnz.value].copy(), val[:nnz.value].copy()
            if nnz.value > cap:                      # grow and retry
                cap = self._sparse_cap = int(min(self.p, max(2 * cap, nnz.value)))
                continue
            self._chk(rc)

    def load_output_dense(self, X_out):
        """Rows EBVs are reported for when they differ from the training rows (Mi.output_genotypes,
        tools4genotypes.jl:290-296).  X_out: n_out x p float32."""
        X_out = np.asarray(X_out)
        if X_out.dtype != np.float32:
            raise TypeError("the HIP path stores Float32 genotypes (double_precision=false)")
        if X_out.ndim != 2:
            raise ValueError("genotype matrix must be 2-D")
        if not X_out.flags.f_contiguous:
            X_out = np.asfortranarray(X_out)
        n_out, p = X_out.shape
        self._chk(self._L.jwas_hip_load_output_dense_f32(self._h, _ptr(X_out), n_out, p, n_out))
        self.n_out = n_out

    def window_sums(self, wptr, idx, val, use_output_rows=False):
        """(sum_i BV_w[i], sum_i BV_w[i]^2) for every window of one marker-effect sample (GWAS.jl:152-165); CSR-like
        description of the nonzero effects: window w = idx/val[wptr[w]:wptr[w+1]]."""
        wptr = np.ascontiguousarray(wptr, dtype=np.int32)
        idx = np.ascontiguousarray(idx, dtype=np.int32)
        val = np.ascontiguousarray(val, dtype=np.float32)
        nwin = wptr.size - 1
        s, q = np.empty(nwin), np.empty(nwin)
        self._chk(self._L.jwas_hip_window_sums(self._h, 1 if use_output_rows else 0, nwin, _ptr(wptr), _ptr(idx), _ptr(val), _ptr(s), _ptr(q)))
        return s, q

    def window_sums2(self, wptr, idx, val1, val2, use_output_rows=False):
        """Two effect vectors over the same markers: (sum1, ss1, sum2, ss2, cross) per window (GWAS.jl:199-217)."""
        wptr = np.ascontiguousarray(wptr, dtype=np.int32)
        idx = np.ascontiguousarray(idx, dtype=np.int32)
        v1 = np.ascontiguousarray(val1, dtype=np.float32)
        v2 = np.ascontiguousarray(val2, dtype=np.float32)
        nwin = wptr.size - 1
        outs = [np.empty(nwin) for _ in range(5)]
        self._chk(self._L.jwas_hip_window_sums2(self._h, 1 if use_output_rows else 0, nwin, _ptr(wptr), _ptr(idx), _ptr(v1), _ptr(v2),
                                                *[_ptr(o) for o in outs]))
        return tuple(outs)

    def mul_alpha_output(self, trait=0):
        """EBV = output_genotypes * alpha (output.jl:281-306)."""
        out = np.empty(getattr(self, "n_out", 0), dtype=np.float32)
        self._chk(self._L.jwas_hip_mul_alpha_output(self._h, int(trait), _ptr(out)))
        return out

    # -- the sweep -------------------------------------------------------------------------------
    # -- marker shards over GPUs (jwas_hip_comm_* / jwas_hip_sweep_sharded) ------------------------------
    @staticmethod
    def comm_unique_id():
        """128 bytes identifying a new RCCL communicator (rank 0 creates it, every rank passes it to comm_init)."""
        buf = (C.c_char * 128)()
        L = _lib.load()
        rc = L.jwas_hip_comm_unique_id(C.cast(buf, C.c_void_p))
        if rc != 0:
            raise JwasHipError(rc, L.jwas_hip_last_error(None).decode())
        return bytes(buf)

    def comm_init(self, unique_id, rank, world):
        buf = C.create_string_buffer(bytes(unique_id), 128)
        self._chk(self._L.jwas_hip_comm_init(self._h, C.cast(buf, C.c_void_p), int(rank), int(world)))
        self._comm = True

    def comm_info(self):
        """(rank, world) of the attached communicator as RCCL reports them (ncclCommUserRank / ncclCommCount)."""
        r, w = C.c_int32(0), C.c_int32(1)
        self._chk(self._L.jwas_hip_comm_info(self._h, C.byref(r), C.byref(w)))
        return r.value, w.value

    def comm_row_shards(self, enable=True):
        """Exact ROW shards (jwas_hip_comm_row_shards): this rank holds a slice of the individuals and all markers; call
        after comm_init (or comm_init_loopback) and before setup_blocks.  sweep() is then the exact chain of the pooled data."""
        self._chk(self._L.jwas_hip_comm_row_shards(self._h, 1 if enable else 0))

    def comm_init_loopback(self, slot, rank, world):
        """Test transport: the ranks are engines of ONE process driven by different host threads (exchange through host memory)."""
        self._chk(self._L.jwas_hip_comm_init_loopback(self._h, int(slot), int(rank), int(world)))

    def comm_destroy(self):
        self._chk(self._L.jwas_hip_comm_destroy(self._h))
        self._comm = False

    def sweep_sharded(self, **params):
        """sweep() on this rank's markers + the on-device reconcile (one RCCL all-reduce of delta r and the packed
        statistics); the returned statistics are the all-rank sums, the residual (get_residual) is the reconciled one."""
        return self.sweep(_sharded=True, **params)

    def sweep(self, *, iteration, seed, vare, var_effect, pi=0.0, pi_classes=None, gamma=BAYESR_GAMMA,
              log_prior_states=None, var_effect_vec=None, var_effect_matrix=None, pi_vec=None, pi_matrix=None, nreps=1,
              marker_offset=0, independent_blocks=False, section_solve=False, group_launch=False, _sharded=False):
        """One marker sweep.  Argument meaning follows BayesABC!/BayesR!/MTBayesABC!:
        vare: residual variance (scalar or t x t); var_effect: marker effect variance (BayesC scalar,
        BayesR sigmaSq, MT t x t); pi: Pr(effect = 0) scalar, or pi_vec per marker (length p, else the
        reference's length error); pi_classes / pi_matrix: BayesR class priors (4 or p x 4)."""
        t = self.ntraits
        P = SweepParams()
        P.method, P.ntraits, P.nreps = self.method, t, int(nreps)
        P.iteration, P.seed, P.marker_offset = int(iteration), int(seed), int(marker_offset)
        P.independent_blocks = 1 if independent_blocks else 0          # BayesABC_block_independent! (BayesABC.jl:190-255)
        P.section_solve = 1 if section_solve else 0                    # Rule T: dense chains as triangular solves (jwas_hip.h)
        P.group_launch = 1 if group_launch else 0                      # grouped launches (setup_groups; jwas_hip.h)
        ve = np.asarray(vare, dtype=np.float32).reshape(-1)
        vg = np.asarray(var_effect, dtype=np.float32).reshape(-1)
        if ve.size != t * t or vg.size != t * t:
            raise ValueError(f"vare / var_effect must have {t}x{t} entries")
        ve64 = np.asarray(vare, dtype=np.float64).reshape(-1)
        vg64 = np.asarray(var_effect, dtype=np.float64).reshape(-1)
        for i in range(t * t):
            P.vare[i] = float(ve[i])
            P.var_effect[i] = float(vg[i])
            P.vare_f64[i] = float(ve64[i])              # (read by Float64 contexts: the variances as the reference holds them there)
            P.var_effect_f64[i] = float(vg64[i])
        P.pi = float(pi) if np.ndim(pi) == 0 else 0.0      # (vector pi: per marker, or per trait for megaBayesABC)
        keep = []
        if self.method in (_lib.MTBAYESB1, _lib.MTBAYESB2, _lib.MEGABAYESB):     # multi-trait BayesA/B: one t x t effect covariance per marker
            if var_effect_matrix is not None:     # (None: the covariances resident on the device -- sample_marker_covariances)
                vm = np.ascontiguousarray(var_effect_matrix, dtype=np.float32)
                if vm.shape != (self.p, t, t):
                    raise ValueError(f"var_effect_matrix must be {self.p} x {t} x {t}")
                keep.append(vm)
                P.var_effect_matrix = vm.ctypes.data_as(C.POINTER(C.c_float))
        if self.method in (_lib.BAYESC, _lib.BAYESB):
            if np.ndim(pi) == 1:
                pi_vec = pi
            if pi_vec is not None:
                pv = np.ascontiguousarray(pi_vec, dtype=np.float64)
                if pv.shape != (self.p,):
                    # bayesabc_pi_vector (BayesABC.jl:16-22)
                    raise ValueError(f"BayesABC pi vector length {pv.size} must match the number of markers ({self.p}).")
                P.pi_vec = pv.ctypes.data_as(C.POINTER(C.c_double))
                keep.append(pv)
            if self.method == _lib.BAYESB:
                if var_effect_vec is None:
                    raise ValueError("BayesB needs per-marker effect variances")
                vv = np.ascontiguousarray(var_effect_vec, dtype=self.dtype)
                if vv.shape != (self.p,):
                    raise ValueError(f"BayesB variance vector must have length {self.p}")
                if self.precision == 64:
                    P.var_effect_vec_f64 = vv.ctypes.data_as(C.POINTER(C.c_double))
                else:
                    P.var_effect_vec = vv.ctypes.data_as(C.POINTER(C.c_float))
                keep.append(vv)
        elif self.method == _lib.BAYESR:
            g = np.asarray(gamma, dtype=np.float64)
            for k in range(4):
                P.gamma[k] = float(g[k])
            pc = pi_classes
            if pc is not None and np.ndim(pc) == 2:
                pi_matrix, pc = pc, None
            if pi_matrix is not None:
                pm = np.ascontiguousarray(pi_matrix, dtype=np.float64)
                # bayesr_validate_priors (BayesR.jl:16-20)
                if pm.shape[0] != self.p:
                    raise ValueError("BayesR per-marker pi must have one row per marker.")
                if pm.shape[1] != 4:
                    raise ValueError("BayesR per-marker pi must have 4 columns.")
                P.pi_matrix = pm.ctypes.data_as(C.POINTER(C.c_double))
                keep.append(pm)
            else:
                pc = np.asarray(pc, dtype=np.float64)
                if pc.shape != (4,):
                    raise ValueError(f"BayesR pi vector length {pc.size} must match the number of mixture classes (4).")
                for k in range(4):
                    P.pi_classes[k] = float(pc[k])
        elif self.method in (_lib.MEGABAYESC, _lib.MEGABAYESB):
            # megaBayesABC! (BayesABC.jl:1-8): one pi per trait (genotypes.pi[i])
            pt = np.asarray(pi, dtype=np.float64).reshape(-1)
            if pt.shape != (t,):
                raise ValueError(f"megaBayesABC needs one pi per trait ({t}), got {pt.size}")
            for k in range(t):
                P.pi_classes[k] = float(pt[k])
        else:
            lp = np.asarray(log_prior_states, dtype=np.float64)
            if lp.ndim == 2:                      # marker-specific joint priors (MarkerSpecificPiPrior, MTBayesABC.jl:22-47)
                if lp.shape != (self.p, 1 << t):
                    raise ValueError(f"marker-specific log_prior_states must be {self.p} x {1 << t}")
                lpm = np.ascontiguousarray(lp)
                keep.append(lpm)
                P.log_prior_states_matrix = lpm.ctypes.data_as(C.POINTER(C.c_double))
            else:
                if lp.shape != (1 << t,):
                    raise ValueError(f"log_prior_states must have {1 << t} entries")
                for k in range(1 << t):
                    P.log_prior_states[k] = float(lp[k])
        self._keep = keep
        S = SweepStats()
        self._chk((self._L.jwas_hip_sweep_sharded if _sharded else self._L.jwas_hip_sweep)(self._h, C.byref(P), C.byref(S)))
        return {
            "sum_delta": np.array(S.sum_delta[:t]),
            "alpha_ss": np.array(S.alpha_ss[:t * t]).reshape(t, t),
            "beta_ss": np.array(S.beta_ss[:t * t]).reshape(t, t),
            "resid_ss": np.array(S.resid_ss[:t * t]).reshape(t, t),
            "resid_sum": np.array(S.resid_sum[:t]),
            "class_counts": np.array(S.class_counts[:4]),
            "bayesr_ssq": S.bayesr_ssq, "bayesr_nnz": S.bayesr_nnz,
            "state_counts": np.array(S.state_counts[:1 << t]),
            "n_events": S.n_events, "sweep_ms": S.sweep_ms,
            "update_kernel_ms": S.update_kernel_ms, "update_kernel_samples": S.update_kernel_samples,
            "update_kernel_bytes": S.update_kernel_bytes, "event_overhead_ms": S.event_overhead_ms,
        }

    # -- multi-trait BayesA/B: per-marker effect covariances drawn on the device --------------------
    def sample_marker_covariances(self, df, scale, *, seed, iteration, marker_offset=0):
        """G_j ~ InverseWishart(df, scale + b_j b_j') for every marker from the current beta (variance_components.jl:181-186,
        df = the reference's df + 1); the next sweep(var_effect_matrix=None) uses them in place."""
        sc = np.ascontiguousarray(scale, dtype=np.float64).reshape(-1)
        if sc.size != self.ntraits * self.ntraits:
            raise ValueError(f"scale must be {self.ntraits} x {self.ntraits}")
        self._chk(self._L.jwas_hip_sample_marker_covariances(self._h, float(df), sc.ctypes.data_as(C.POINTER(C.c_double)), int(seed),
                                                             int(iteration), int(marker_offset)))

    def marker_covariances(self):
        out = np.empty((self.p, self.ntraits, self.ntraits), dtype=np.float32)
        self._chk(self._L.jwas_hip_get_marker_covariances(self._h, _ptr(out)))
        return out

    # -- posterior accumulators --------------------------------------------------------------------
    def accumulate(self, nsamples):
        self._chk(self._L.jwas_hip_accumulate(self._h, float(nsamples)))

    def posterior(self, trait=0):
        ma = np.empty(self.p, dtype=self.dtype)
        ma2 = np.empty(self.p, dtype=self.dtype)
        md = np.empty(self.p, dtype=self.dtype)
        self._chk(self._f("jwas_hip_get_posterior")(self._h, int(trait), _ptr(ma), _ptr(ma2), _ptr(md)))
        return ma, ma2, md
