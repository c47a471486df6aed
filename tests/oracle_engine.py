"""OracleEngine: the sweep-engine protocol of jwas.jl_amd.engine.HipEngine implemented on the CPU
oracle (oracle/jwas_oracle.c).  TEST INFRASTRUCTURE: lives under tests/, never imported by the
package.  It lets the host MCMC loop run on CPU so GPU chains can be compared end to end.
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "oracle"))
import oracle as O  # noqa: E402

BAYESC, BAYESB, BAYESR, MTBAYESC1, MTBAYESC2, MEGABAYESC, MTBAYESB1, MTBAYESB2, MEGABAYESB = 0, 1, 2, 3, 4, 5, 6, 7, 8
METHOD_CODES = {"BayesC": BAYESC, "BayesB": BAYESB, "BayesA": BAYESB, "BayesR": BAYESR, "MTBayesC": MTBAYESC1,
                "MTBayesC_II": MTBAYESC2, "MegaBayesC": MEGABAYESC, "MTBayesB": MTBAYESB1, "MTBayesB_II": MTBAYESB2, "MegaBayesB": MEGABAYESB}


class OracleEngine:
    """form='lookahead' mirrors the device schedule exactly (exact block form, one-block lookahead,
    same block size); form='block' is the plain block restatement (BayesABC.jl:118-188);
    form='dense' is the literal non-block restatement (BayesABC.jl:60-80)."""

    def __init__(self, form="block", acc=O.ACC_F64):
        self.form, self.acc = form, acc
        self.n = self.p = 0
        self.method = None
        self.ntraits = 0
        self.block_size = 0

    def close(self):
        pass

    def load_dense(self, X):
        X = np.asfortranarray(np.asarray(X, dtype=np.float32))
        self.X = X
        self.n, self.p = X.shape

    def set_weights(self, rinv):
        """Residual weights R^-1 (jwas_hip_set_weights); like the device, drops the block configurations."""
        self._rinv = None if rinv is None else np.ascontiguousarray(rinv, dtype=np.float32)
        if self._rinv is not None and np.all(self._rinv == 1):
            self._rinv = None
        self.block_size = 0

    def _w(self):
        O.set_weights(getattr(self, "_rinv", None))

    def load_jgb2(self, path):
        """CPU stand-in for jwas_hip_load_jgb2: decode the packed backend (decode_marker!) and keep it dense."""
        from jwas_jl_amd import streaming as S
        self.load_dense(S.decode_markers(S.load_streaming_backend(path)))

    def setup_blocks(self, block_size=256, gram_mode="f64"):
        self.block_size = int(block_size)
        self._w()
        self._xpx = O.xpx(self.X, self.acc)
        self._bs = O.block_starts_for(self.p, self.block_size)
        self._grams = O.grams_for(self.X, self._bs, self.acc)
        self._sets = {self.block_size: (self._bs, self._grams)}
        self._groups = {}
        O.set_weights(None)

    def setup_blocks_explicit(self, starts, gram_mode="f64"):
        self._w()
        self._xpx = O.xpx(self.X, self.acc)
        self._bs = np.ascontiguousarray(starts, dtype=np.int64)
        self._grams = O.grams_for(self.X, self._bs, self.acc)
        sizes = np.diff(np.append(self._bs, self.p))
        self.block_size = int(sizes.max())
        self._sets = {self.block_size: (self._bs, self._grams)}
        self._groups = {}
        O.set_weights(None)

    def add_block_size(self, block_size, gram_mode="f64"):
        self._w()
        bs = O.block_starts_for(self.p, int(block_size))
        self._sets[int(block_size)] = (bs, O.grams_for(self.X, bs, self.acc))
        O.set_weights(None)

    def select_block_size(self, block_size):
        self._bs, self._grams = self._sets[int(block_size)]
        self.block_size = int(block_size)

    @property
    def nblocks(self):
        return len(self._bs)

    def block_starts(self):
        return self._bs

    def xpx(self):
        return self._xpx.copy()

    def grams_packed(self):
        return self._grams

    def set_grams_packed(self, g):
        self._grams = np.ascontiguousarray(g, dtype=np.float32)

    def init_state(self, method, ntraits=1):
        self.method = METHOD_CODES[method] if isinstance(method, str) else int(method)
        self.ntraits = int(ntraits)
        t, p = self.ntraits, self.p
        self.alpha = np.zeros((t, p), dtype=np.float32)
        self.beta = np.zeros((t, p), dtype=np.float32)
        self.delta = np.zeros((t, p), dtype=np.int32 if self.method == BAYESR else np.float32)
        self.r = np.zeros((t, self.n), dtype=np.float32)
        self.mean_a = np.zeros((t, p), dtype=np.float32)
        self.mean_a2 = np.zeros((t, p), dtype=np.float32)
        self.mean_d = np.zeros((t, p), dtype=np.float32)

    def set_state(self, trait=0, alpha=None, beta=None, delta=None):
        if alpha is not None:
            self.alpha[trait] = alpha
        if beta is not None:
            self.beta[trait] = beta
        if delta is not None:
            self.delta[trait] = delta

    def get_state(self, trait=0):
        return self.alpha[trait].copy(), self.beta[trait].copy(), self.delta[trait].copy()

    def set_residual(self, r, trait=0):
        self.r[trait] = r

    def get_residual(self, trait=0):
        return self.r[trait].copy()

    def sub_xalpha(self, trait=0):
        O.residual_minus_xalpha(self.X, self.alpha[trait], self.r[trait])

    def mul_alpha(self, trait=0):
        return (self.X.astype(np.float64) @ self.alpha[trait].astype(np.float64)).astype(np.float32)

    def window_sums(self, wptr, idx, val, use_output_rows=False):
        X = (self.X_out if use_output_rows else self.X).astype(np.float64)
        idx = np.asarray(idx, dtype=np.int64); val = np.asarray(val, dtype=np.float64)
        s, q = np.zeros(len(wptr) - 1), np.zeros(len(wptr) - 1)
        for w in range(len(wptr) - 1):
            bv = X[:, idx[wptr[w]:wptr[w + 1]]] @ val[wptr[w]:wptr[w + 1]]
            s[w], q[w] = bv.sum(), (bv * bv).sum()
        return s, q

    def window_sums2(self, wptr, idx, val1, val2, use_output_rows=False):
        X = (self.X_out if use_output_rows else self.X).astype(np.float64)
        idx = np.asarray(idx, dtype=np.int64)
        v1, v2 = np.asarray(val1, dtype=np.float64), np.asarray(val2, dtype=np.float64)
        outs = [np.zeros(len(wptr) - 1) for _ in range(5)]
        for w in range(len(wptr) - 1):
            cols = X[:, idx[wptr[w]:wptr[w + 1]]]
            b1, b2 = cols @ v1[wptr[w]:wptr[w + 1]], cols @ v2[wptr[w]:wptr[w + 1]]
            for o, v in zip(outs, (b1.sum(), (b1 * b1).sum(), b2.sum(), (b2 * b2).sum(), (b1 * b2).sum())):
                o[w] = v
        return tuple(outs)

    def load_output_dense(self, X_out):
        self.X_out = np.asarray(X_out, dtype=np.float32)

    def mul_alpha_output(self, trait=0):
        return (self.X_out.astype(np.float64) @ self.alpha[trait].astype(np.float64)).astype(np.float32)

    def set_packed_source(self, codes, means, centered=True):
        """Block right-hand sides in the 2-bit packed update role's own order (oracle.set_packed_source) for THIS engine's matrix:
        codes n x p (0..3, 3 = missing) whose decoded form was loaded with load_dense.  codes=None: off."""
        if codes is None:
            O.set_packed_source(None, None, True, None)
        else:
            O.set_packed_source(codes, means, centered, self.X)

    def setup_groups(self, blocks_per_launch, gram_mode="f64"):
        """jwas_hip_setup_groups: the sweeps that pass group_launch=True run the grouped lookahead (oracle: la_group_sweep)."""
        self._groups[self.block_size] = int(blocks_per_launch)

    def blocks_per_launch(self, block_size=None):
        return self._groups.get(self.block_size if block_size is None else int(block_size), 0)

    def sweep(self, *, iteration, seed, vare, var_effect, pi=0.0, pi_classes=None, gamma=O.GAMMA,
              log_prior_states=None, var_effect_vec=None, var_effect_matrix=None, pi_vec=None, pi_matrix=None, nreps=1,
              marker_offset=0, independent_blocks=False, section_solve=False, group_launch=False):
        t = self.ntraits
        blk = (dict(block_starts=self._bs, grams=self._grams, nreps=nreps, lookahead=(self.form == "lookahead"),
                    independent=bool(independent_blocks))
               if self.form in ("block", "lookahead") else {})
        if independent_blocks and not blk:
            raise ValueError("independent blocks need a block form")
        a_before = self.alpha.copy()
        self._w()
        if self.method in (MTBAYESB1, MTBAYESB2, MEGABAYESB):     # multi-trait BayesA/B: one effect covariance per marker
            if var_effect_matrix is None:             # (None: the resident ones -- an earlier sweep's or sample_marker_covariances')
                var_effect_matrix = self._var_mat
            self._var_mat = np.ascontiguousarray(var_effect_matrix, dtype=np.float32)
            vm = self._var_mat
            assert vm.shape == (self.p, t, t)
            O.set_var_effect_matrix(vm)
            var_effect = np.eye(t, dtype=np.float32)  # (unused)
        O.set_section_solve(bool(section_solve) and self.form == "lookahead" and not independent_blocks)      # Rule T (the device's section_solve)
        # grouped launches (the device's group_launch after setup_groups): single trait, one pass, uniform blocks, not a uniform pi = 0
        grouped = (bool(group_launch) and self.form == "lookahead" and not independent_blocks and self._groups.get(self.block_size, 0) >= 2 and
                   self.method in (BAYESC, BAYESB, BAYESR) and nreps == 1 and
                   not (self.method in (BAYESC, BAYESB) and pi_vec is None and np.ndim(pi) == 0 and float(pi) == 0.0))
        O.set_lookahead_group(self._groups[self.block_size] if grouped else 1)
        try:
            self._sweep_inner(t, blk, iteration, seed, vare, var_effect, pi, pi_classes, gamma, log_prior_states,
                              var_effect_vec, pi_vec, pi_matrix, marker_offset)
        finally:
            O.set_section_solve(False)
            O.set_lookahead_group(1)
            O.set_weights(None)
            O.set_var_effect_matrix(None)
        return self._stats(a_before, gamma)

    def _sweep_inner(self, t, blk, iteration, seed, vare, var_effect, pi, pi_classes, gamma, log_prior_states,
                     var_effect_vec, pi_vec, pi_matrix, marker_offset):
        if self.method in (BAYESC, BAYESB):
            if np.ndim(pi) == 1:
                pi_vec = pi
            pv = pi_vec if pi_vec is not None else float(pi)
            ve = var_effect_vec if self.method == BAYESB else float(np.asarray(var_effect).reshape(-1)[0])
            O.bayesabc_sweep(self.X, self._xpx, self.r[0], self.alpha[0], self.beta[0], self.delta[0],
                             float(np.asarray(vare).reshape(-1)[0]), ve, pv, seed, iteration,
                             marker0=marker_offset, acc=self.acc, **blk)
        elif self.method == BAYESR:
            pc = pi_matrix if pi_matrix is not None else pi_classes
            O.bayesr_sweep(self.X, self._xpx, self.r[0], self.alpha[0], self.delta[0],
                           float(np.asarray(vare).reshape(-1)[0]), float(np.asarray(var_effect).reshape(-1)[0]),
                           pc, seed, iteration, gamma=gamma, marker0=marker_offset, acc=self.acc, **blk)
        else:
            kind = {MTBAYESC1: O.MT_SAMPLER_I, MTBAYESB1: O.MT_SAMPLER_I, MTBAYESC2: O.MT_SAMPLER_II, MTBAYESB2: O.MT_SAMPLER_II, MEGABAYESC: O.MT_MEGA, MEGABAYESB: O.MT_MEGA}[self.method]
            prior = np.asarray(pi, dtype=np.float64).reshape(-1) if self.method in (MEGABAYESC, MEGABAYESB) else log_prior_states
            O.mt_sweep(kind, self.X, self._xpx, self.r, self.alpha, self.beta, self.delta,
                       np.asarray(vare, dtype=np.float32).reshape(t, t),
                       np.asarray(var_effect, dtype=np.float32).reshape(t, t),
                       prior, seed, iteration, marker0=marker_offset, acc=self.acc, **blk)

    def _stats(self, a_before, gamma):
        t = self.ntraits
        a64, b64, r64 = self.alpha.astype(np.float64), self.beta.astype(np.float64), self.r.astype(np.float64)
        w64 = np.ones(self.n) if getattr(self, "_rinv", None) is None else self._rinv.astype(np.float64)
        out = {
            "alpha_ss": a64 @ a64.T, "beta_ss": b64 @ b64.T, "resid_ss": (r64 * w64) @ r64.T,      # r'R^-1 r
            "resid_sum": (r64 * w64).sum(axis=1), "n_events": float(np.any(a_before != self.alpha, axis=0).sum()),
            "sweep_ms": 0.0, "class_counts": np.zeros(4), "bayesr_ssq": 0.0, "bayesr_nnz": 0.0,
            "sum_delta": np.zeros(t), "state_counts": np.zeros(1 << t),
        }
        if self.method == BAYESR:
            d = self.delta[0]
            out["class_counts"] = np.array([(d == k + 1).sum() for k in range(4)], dtype=np.float64)
            ssq, nnz = O.bayesr_sigma_suffstats(self.alpha[0], d, gamma)
            out["bayesr_ssq"], out["bayesr_nnz"] = ssq, float(nnz)
        else:
            d = self.delta
            out["sum_delta"] = d.astype(np.float64).sum(axis=1)
            state = np.zeros(self.p, dtype=np.int64)
            for k in range(t):
                state |= (d[k] != 0).astype(np.int64) << k
            out["state_counts"] = np.bincount(state, minlength=1 << t).astype(np.float64)
        return out

    def sample_marker_covariances(self, df, scale, *, seed, iteration, marker_offset=0):
        """The oracle's restatement of jwas_hip_sample_marker_covariances (same counter RNG, same operations)."""
        self._var_mat = O.sample_marker_covariances(self.beta, df, scale, seed, iteration, marker_offset,
                                                    diagonal=(self.method == MEGABAYESB))

    def marker_covariances(self):
        return self._var_mat.copy()

    def accumulate(self, nsamples):
        for k in range(self.ntraits):
            O.accumulate(self.alpha[k], self.delta[k], float(nsamples), self.mean_a[k], self.mean_a2[k], self.mean_d[k])

    def posterior(self, trait=0):
        return self.mean_a[trait].copy(), self.mean_a2[trait].copy(), self.mean_d[trait].copy()


class OracleEngine64:
    """The sweep-engine protocol on the Float64 oracle (oracle/jwas_oracle_f64.c): runMCMC(double_precision=true) on the CPU.
    The literal non-block chain (block form with repetitions for single-trait BayesA/B/C when nreps != 1)."""
    precision = 64
    dtype = np.float64

    def __init__(self):
        self.n = self.p = 0
        self.method = None
        self.ntraits = 0
        self.block_size = 0

    def close(self):
        pass

    def load_dense(self, X):
        self.X = np.asfortranarray(np.asarray(X, dtype=np.float64))
        self.n, self.p = self.X.shape

    def setup_blocks(self, block_size=128, gram_mode="f64"):
        self.block_size = int(block_size)
        self._starts = None
        self._xpx = O.xpx64_w(self.X, getattr(self, "_w", None))

    def setup_blocks_explicit(self, starts, gram_mode="f64"):
        self._starts = np.asarray(starts, dtype=np.int64)
        sizes = np.diff(np.append(self._starts, self.p))
        self.block_size = int(sizes.max())
        self._xpx = O.xpx64_w(self.X, getattr(self, "_w", None))

    def set_weights(self, rinv):
        self._w = None if rinv is None else np.asarray(rinv, dtype=np.float64)      # (jwas_hip_set_weights_f64: the Float64 values as they are)
        if getattr(self, "_xpx", None) is not None:
            self._xpx = O.xpx64_w(self.X, self._w)

    def block_starts(self):
        if getattr(self, "_starts", None) is not None:
            return self._starts.copy()
        return np.arange(0, self.p, self.block_size, dtype=np.int64)

    def xpx(self):
        return self._xpx.copy()

    def init_state(self, method, ntraits=1):
        self.method = METHOD_CODES[method] if isinstance(method, str) else int(method)
        assert self.method in (BAYESC, BAYESB, BAYESR, MTBAYESC1)
        self.ntraits = t = int(ntraits)
        self.alpha = np.zeros((t, self.p)); self.beta = np.zeros((t, self.p))
        self.delta = np.zeros(self.p, dtype=np.int32)[None] if self.method == BAYESR else np.zeros((t, self.p))
        self.r = np.zeros((t, self.n))
        self.mean_a = np.zeros((t, self.p)); self.mean_a2 = np.zeros((t, self.p)); self.mean_d = np.zeros((t, self.p))

    def set_state(self, trait=0, alpha=None, beta=None, delta=None):
        if alpha is not None: self.alpha[trait] = alpha
        if beta is not None: self.beta[trait] = beta
        if delta is not None: self.delta[trait] = delta

    def get_state(self, trait=0):
        return self.alpha[trait].copy(), self.beta[trait].copy(), self.delta[trait].copy()

    def set_residual(self, r, trait=0):
        self.r[trait] = np.asarray(r, dtype=np.float64)

    def get_residual(self, trait=0):
        return self.r[trait].copy()

    def sub_xalpha(self, trait=0):
        self.r[trait] -= self.X @ self.alpha[trait]

    def mul_alpha(self, trait=0):
        return self.X @ self.alpha[trait]

    def sweep(self, *, iteration, seed, vare, var_effect, pi=0.0, pi_classes=None, gamma=O.GAMMA, log_prior_states=None,
              var_effect_vec=None, pi_vec=None, pi_matrix=None, nreps=1, marker_offset=0, independent_blocks=False, **_):
        t = self.ntraits
        a_before = self.alpha.copy()
        w = getattr(self, "_w", None)
        general = independent_blocks or w is not None or getattr(self, "_starts", None) is not None
        if self.method in (BAYESC, BAYESB):
            if np.ndim(pi) == 1:
                pi_vec = pi
            pv = pi_vec if pi_vec is not None else float(pi)
            ve = var_effect_vec if self.method == BAYESB else float(np.asarray(var_effect).reshape(-1)[0])
            if general:
                O.bayesabc_block_sweep64_ex(self.X, self._xpx, self.r[0], self.alpha[0], self.beta[0], self.delta[0],
                                            float(np.asarray(vare).reshape(-1)[0]), ve, pv, seed, iteration, self.block_starts(),
                                            nreps=nreps, independent=independent_blocks, w=w, marker0=marker_offset)
            else:
                O.bayesabc_sweep64(self.X, self._xpx, self.r[0], self.alpha[0], self.beta[0], self.delta[0],
                                   float(np.asarray(vare).reshape(-1)[0]), ve, pv, seed, iteration, marker0=marker_offset,
                                   block_size=0 if nreps == 1 else self.block_size, nreps=nreps)
        elif self.method == BAYESR:
            assert nreps == 1 and not general
            pc = pi_matrix if pi_matrix is not None else pi_classes
            O.bayesr_sweep64(self.X, self._xpx, self.r[0], self.alpha[0], self.delta[0], float(np.asarray(vare).reshape(-1)[0]),
                             float(np.asarray(var_effect).reshape(-1)[0]), pc, seed, iteration, gamma=gamma, marker0=marker_offset)
        else:
            assert nreps == 1 and not general
            O.mt1_sweep64(self.X, self._xpx, self.r, self.alpha, self.beta, self.delta, np.asarray(vare, dtype=np.float64).reshape(t, t),
                          np.asarray(var_effect, dtype=np.float64).reshape(t, t), log_prior_states, seed, iteration, marker0=marker_offset)
        ww = np.ones(self.n) if w is None else w
        out = {"alpha_ss": self.alpha @ self.alpha.T, "beta_ss": self.beta @ self.beta.T, "resid_ss": (self.r * ww) @ self.r.T,
               "resid_sum": (self.r * ww).sum(axis=1), "n_events": float(np.any(a_before != self.alpha, axis=0).sum()), "sweep_ms": 0.0,
               "class_counts": np.zeros(4), "bayesr_ssq": 0.0, "bayesr_nnz": 0.0, "sum_delta": np.zeros(t), "state_counts": np.zeros(1 << t)}
        if self.method == BAYESR:
            d = self.delta[0]
            out["class_counts"] = np.array([(d == k + 1).sum() for k in range(4)], dtype=np.float64)
            nz = d > 1
            out["bayesr_ssq"] = float((self.alpha[0][nz] ** 2 / np.asarray(gamma)[d[nz] - 1]).sum())
            out["bayesr_nnz"] = float(nz.sum())
        else:
            out["sum_delta"] = self.delta.sum(axis=1)
            state = np.zeros(self.p, dtype=np.int64)
            for k in range(t):
                state |= (self.delta[k] != 0).astype(np.int64) << k
            out["state_counts"] = np.bincount(state, minlength=1 << t).astype(np.float64)
        return out

    def accumulate(self, nsamples):
        k = float(nsamples)
        d = (self.delta > 1).astype(np.float64) if self.method == BAYESR else self.delta
        self.mean_a += (self.alpha - self.mean_a) / k
        self.mean_a2 += (self.alpha ** 2 - self.mean_a2) / k
        self.mean_d += (d - self.mean_d) / k

    def posterior(self, trait=0):
        return self.mean_a[trait].copy(), self.mean_a2[trait].copy(), self.mean_d[trait].copy()
