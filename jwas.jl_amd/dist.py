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
    gives  r = r_snapshot + sum_g dr_g.  With the nccl backend and a HipEngine all of that runs INSIDE the library
    (jwas_hip_comm_init / jwas_hip_sweep_sharded: HIP kernels + ncclAllReduce over xGMI on the context's stream) and this
    class only hands the RCCL id around; the numpy form below serves the CPU tests (gloo) and non-HIP engines.
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


def _attach_library_comm(engine, rank, world, group, dev):
    """Create the library's RCCL communicator on every rank, or on none.  ncclCommInitRank is collective, so the ranks
    first agree on a NON-collective probe (can this rank bind librccl at all: jwas_hip_comm_unique_id does the dlopen and
    a local ncclGetUniqueId) and only enter comm_init when every rank can; a rank that fails afterwards raises (a hang
    would be the alternative).  JWAS_DIST_TORCH_RECONCILE=1 forces the torch.distributed reconcile."""
    import os
    import torch
    import torch.distributed as dist
    ok = 0
    if os.environ.get("JWAS_DIST_TORCH_RECONCILE", "0") == "0":
        try:
            engine.comm_unique_id()
            ok = 1
        except Exception as ex:                  # noqa: BLE001  (e.g. librccl.so not loadable)
            print(f"[jwas] library RCCL communicator not available on rank {rank} ({ex}); "
                  "falling back to the torch.distributed reconcile")
    flag = torch.tensor([ok], device=dev)
    dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=group)
    if not int(flag.item()):
        return False
    box = [engine.comm_unique_id() if rank == 0 else None]
    dist.broadcast_object_list(box, src=0, group=group)
    engine.comm_init(box[0], rank, world)
    r, w = engine.comm_info()
    if (r, w) != (rank, world):
        raise RuntimeError(f"RCCL communicator reports rank {r} of {w}, expected rank {rank} of {world}")
    return True


class MarkerShard:
    """Wraps one rank's sweep engine (its marker shard already loaded) and reconciles after a sweep."""

    def __init__(self, engine, lo, hi, rank=0, world=1, group=None, force_collective=False):
        self.engine, self.lo, self.hi = engine, int(lo), int(hi)
        self.rank, self.world, self.group = int(rank), int(world), group
        self._dist = None
        self._dev = None
        self._coll = world > 1 or force_collective      # force_collective: exercise the exchange with one rank (tests)
        self._stream_set = False
        self._lib_comm = False
        if self._coll:
            import torch
            import torch.distributed as dist
            self._torch, self._dist = torch, dist
            if dist.get_backend(group) == "nccl":
                self._dev = torch.device("cuda", torch.cuda.current_device())
                if hasattr(engine, "comm_init"):
                    # the reconcile runs inside the library (jwas_hip_sweep_sharded: pack kernel, ncclAllReduce on the
                    # context's stream, apply kernel); torch.distributed only hands the 128-byte RCCL id around
                    self._lib_comm = _attach_library_comm(engine, self.rank, self.world, group, self._dev)

    def allreduce_sum(self, arr):
        """Sum a numpy array over ranks (deterministic: every rank receives the same bits)."""
        if not self._coll:
            return arr
        t = self._torch.from_numpy(np.ascontiguousarray(arr))
        if self._dev is not None:
            t = t.to(self._dev)
        self._dist.all_reduce(t, op=self._dist.ReduceOp.SUM, group=self.group)
        return t.cpu().numpy()

    def sweep_resident(self, **params):
        """One sweep + reconcile with the residual RESIDENT on the device: the engine's current residual is the replicated
        snapshot, and on return it holds the reconciled one (jwas_hip_sweep_sharded does all of it on the context's
        stream).  Returns the all-rank statistics (resid_sum / resid_ss of the reconciled residual included), which is all
        a host with an intercept-only location step needs (engine.residual_add_scalar applies its correction)."""
        eng = self.engine
        if self._lib_comm:
            return eng.sweep_sharded(marker_offset=self.lo, **params)
        if not self._coll:
            return eng.sweep(marker_offset=self.lo, **params)
        snap = np.stack([eng.get_residual(k) for k in range(eng.ntraits)])      # (gloo / torch reconcile: through the host)
        return self.sweep(snap, **params)[1]

    def comm_world(self):
        """Number of ranks the transport itself reports (ncclCommCount for the library communicator)."""
        if self._lib_comm:
            return self.engine.comm_info()[1]
        return self._dist.get_world_size(self.group) if self._coll else 1

    def sweep(self, r_snapshot, **params):
        """r_snapshot: t x n float32 (replicated).  Returns (r_new t x n, stats) after reconcile.

        Reconcile:  r_new = fl32( r_snapshot + sum_g (r_local_g - r_snapshot) ),  the differences and their sum in
        float64 (exact for one rank; rank-order independent up to the fixed reduction order of the collective), in ONE
        all-reduce that also carries the packed O(p) statistics."""
        eng = self.engine
        t = r_snapshot.shape[0]
        if self._lib_comm:
            for k in range(t):
                eng.set_residual(r_snapshot[k], k)
            st = eng.sweep_sharded(marker_offset=self.lo, **params)      # all-rank statistics, reconciled residual
            return np.stack([eng.get_residual(k) for k in range(t)]), st
        if self._dev is not None:
            return self._sweep_device(r_snapshot, **params)
        for k in range(t):
            eng.set_residual(r_snapshot[k], k)
        st = eng.sweep(marker_offset=self.lo, **params)
        if not self._coll:
            r_new = np.stack([eng.get_residual(k) for k in range(t)])
            return r_new, st
        snap64 = r_snapshot.astype(np.float64)
        delta = np.stack([eng.get_residual(k) for k in range(t)]).astype(np.float64) - snap64          # dr_g
        packed = np.concatenate([np.atleast_1d(np.asarray(st[k], dtype=np.float64)).ravel() for k in _PACK])
        total = self.allreduce_sum(np.concatenate([delta.ravel(), packed]))
        r_new = (snap64 + total[:delta.size].reshape(delta.shape)).astype(np.float32)
        self._unpack(st, total[delta.size:], r_new)
        for k in range(t):
            eng.set_residual(r_new[k], k)
        return r_new, st

    @staticmethod
    def _unpack(st, packed, r_new):
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

    def _sweep_device(self, r_snapshot, **params):
        """nccl backend: the residual stays on the GPU between the sweep and the collective (device-to-device copies in
        and out of the library on torch's stream, RCCL all-reduce over xGMI, one host copy of the reconciled residual)."""
        torch, eng = self._torch, self.engine
        t, n = r_snapshot.shape
        if not self._stream_set:
            eng.set_stream(torch.cuda.current_stream().cuda_stream)     # order the library's work with torch's
            self._stream_set = True
        snap = torch.from_numpy(np.ascontiguousarray(r_snapshot, dtype=np.float32)).to(self._dev)
        for k in range(t):
            eng.residual_from_dev(snap[k].data_ptr(), k)
        st = eng.sweep(marker_offset=self.lo, **params)
        if not self._coll:
            r_new = np.stack([eng.get_residual(k) for k in range(t)])
            return r_new, st
        rloc = torch.empty((t, n), dtype=torch.float32, device=self._dev)
        for k in range(t):
            eng.residual_to_dev(rloc[k].data_ptr(), k)
        packed = np.concatenate([np.atleast_1d(np.asarray(st[k], dtype=np.float64)).ravel() for k in _PACK])
        buf = torch.cat([(rloc.double() - snap.double()).reshape(-1), torch.from_numpy(packed).to(self._dev)])
        self._dist.all_reduce(buf, op=self._dist.ReduceOp.SUM, group=self.group)
        r_dev = (snap.double() + buf[:t * n].reshape(t, n)).float().contiguous()
        for k in range(t):
            eng.residual_from_dev(r_dev[k].data_ptr(), k)
        r_new = r_dev.cpu().numpy()
        self._unpack(st, buf[t * n:].cpu().numpy(), r_new)
        return r_new, st


class RowShard:
    """Exact row shards (the "exact alternative"; jwas_hip_comm_row_shards): rank g holds a slice
    of the INDIVIDUALS and all markers.  x'x, the block Grams and the cross-Grams are summed over the ranks at setup, every
    block's partial right-hand side X_b'r is summed over the ranks (one small ncclAllReduce per block launch on the
    context's stream) before its sampler runs -- replicated, on identical inputs -- so every rank holds the same effects
    and its own slice of the residual: engine.sweep IS the exact chain of the pooled data.

    Create it AFTER the rank's rows are loaded and BEFORE setup_blocks; every rank needs the same number of 256-row
    groups (pad the shorter slices with zero rows).  torch.distributed only hands the 128-byte RCCL id around."""

    def __init__(self, engine, rank, world, group=None):
        import torch
        import torch.distributed as dist
        self.engine, self.rank, self.world, self.group = engine, int(rank), int(world), group
        self._torch, self._dist = torch, dist
        self._dev = None
        if dist.is_initialized():
            if dist.get_backend(group) != "nccl":
                raise RuntimeError("row shards exchange every block's right-hand side through RCCL on the device: they need the nccl backend")
            self._dev = torch.device("cuda", torch.cuda.current_device())
            box = [engine.comm_unique_id() if self.rank == 0 else None]
            dist.broadcast_object_list(box, src=0, group=group)
            uid = box[0]
        else:
            if self.world != 1:
                raise RuntimeError("row shards over more than one rank need an initialised torch.distributed process group")
            uid = engine.comm_unique_id()
        engine.comm_init(uid, self.rank, self.world)
        r, w = engine.comm_info()
        if (r, w) != (self.rank, self.world):
            raise RuntimeError(f"RCCL communicator reports rank {r} of {w}, expected rank {self.rank} of {self.world}")
        engine.comm_row_shards(True)

    def comm_world(self):
        return self.engine.comm_info()[1]

    def allreduce_sum(self, arr):
        if self._dev is None:
            return arr
        t = self._torch.from_numpy(np.ascontiguousarray(arr)).to(self._dev)
        self._dist.all_reduce(t, op=self._dist.ReduceOp.SUM, group=self.group)
        return t.cpu().numpy()

    def sweep_resident(self, **params):
        """engine.sweep on the pooled data: marker statistics replicated, r'r and sum(r) pooled over the ranks."""
        return self.engine.sweep(**params)
