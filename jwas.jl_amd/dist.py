"""Marker-shard parallelism over the GPUs of one node.

The single-site chain is sequential in the marker index, so it does not shard exactly.  What the
reference itself ships for parallel blocks is `independent_blocks=true`
(BayesABC.jl:190-255): every block starts from the same residual snapshot, runs its own chain, and
the residual is reconciled once per sweep by summing the blocks' X_b * delta_alpha_b
(BayesABC.jl:251-253).  This module is that mode with one "block" per GPU:

    rank g owns marker columns [lo_g, hi_g) (all n rows) and the matching alpha/beta/delta/x'x/Gram;
    the residual r (n*t floats) is REPLICATED;
    per sweep: every rank sweeps its own markers exactly (blocked single-site chain, one pass)
    starting from the same r, forms  dr_g = r_local - r_snapshot,  and one all-reduce(sum) of dr
    (RCCL over xGMI with the nccl backend; gloo on CPU for tests) gives  r = r_snapshot + sum_g dr_g.
    The O(p) reductions the host draws need (sum delta, alpha'alpha, class/state counts) are summed in
    the same exchange.

It is an approximation (exact iff X_g' X_h = 0 for g != h, docs/src/manual/block_bayesc.md:116-134)
and is labelled as such; with world_size == 1 it is the exact chain and no collective is issued.
Random draws are keyed by the GLOBAL marker index (marker_offset), so which GPU owns a marker
never changes its draws.
"""
import numpy as np

_PACK = ("sum_delta", "alpha_ss", "beta_ss", "class_counts", "bayesr_ssq", "bayesr_nnz", "state_counts", "n_events")


def shard_range(p_total, rank, world, align=1):
    """Contiguous, balanced marker range of `rank`; shard boundaries are multiples of `align`."""
    units = (p_total + align - 1) // align
    lo_u = (units * rank) // world
    hi_u = (units * (rank + 1)) // world
    return min(lo_u * align, p_total), min(hi_u * align, p_total)


class MarkerShard:
    """Wraps one rank's sweep engine (its marker shard already loaded) and reconciles after a sweep."""

    def __init__(self, engine, lo, hi, rank=0, world=1, group=None, force_collective=False):
        self.engine, self.lo, self.hi = engine, int(lo), int(hi)
        self.rank, self.world, self.group = int(rank), int(world), group
        self._dist = None
        self._dev = None
        self._coll = world > 1 or force_collective      # force_collective: exercise the exchange with one rank (tests)
        if self._coll:
            import torch
            import torch.distributed as dist
            self._torch, self._dist = torch, dist
            if dist.get_backend(group) == "nccl":
                self._dev = torch.device("cuda", torch.cuda.current_device())

    def allreduce_sum(self, arr):
        """Sum a numpy array over ranks (deterministic: every rank receives the same bits)."""
        if not self._coll:
            return arr
        t = self._torch.from_numpy(np.ascontiguousarray(arr))
        if self._dev is not None:
            t = t.to(self._dev)
        self._dist.all_reduce(t, op=self._dist.ReduceOp.SUM, group=self.group)
        return t.cpu().numpy()

    def sweep(self, r_snapshot, **params):
        """r_snapshot: t x n float32 (replicated).  Returns (r_new t x n, stats) after reconcile."""
        eng = self.engine
        t = r_snapshot.shape[0]
        for k in range(t):
            eng.set_residual(r_snapshot[k], k)
        st = eng.sweep(marker_offset=self.lo, **params)
        if not self._coll:
            r_new = np.stack([eng.get_residual(k) for k in range(t)])
            return r_new, st
        delta = np.stack([eng.get_residual(k) for k in range(t)]) - r_snapshot        # dr_g (fp32)
        packed = np.concatenate([np.atleast_1d(np.asarray(st[k], dtype=np.float64)).ravel() for k in _PACK])
        delta = self.allreduce_sum(delta.astype(np.float32))
        packed = self.allreduce_sum(packed)
        r_new = (r_snapshot + delta).astype(np.float32)
        off = 0
        for k in _PACK:
            shape = np.shape(st[k])
            size = int(np.prod(shape)) if shape else 1
            v = packed[off:off + size]
            st[k] = v.reshape(shape) if shape else float(v[0])
            off += size
        r64 = r_new.astype(np.float64)
        st["resid_ss"] = r64 @ r64.T
        st["resid_sum"] = r64.sum(axis=1)
        for k in range(t):
            eng.set_residual(r_new[k], k)
        return r_new, st
