"""Worker for tests/test_dist_gloo.py: one rank of a 2-process gloo group running the marker-shard
sweep (jwas.jl_amd/dist.py) on the CPU oracle engine; writes its result for the parent to check."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
for p in (os.path.dirname(HERE), os.path.join(os.path.dirname(HERE), "oracle"), HERE):
    if p not in sys.path:
        sys.path.insert(0, p)

import torch.distributed as dist  # noqa: E402

from conftest import make_dataset  # noqa: E402
from oracle_engine import OracleEngine  # noqa: E402
from jwas_jl_amd.dist import MarkerShard, shard_range  # noqa: E402


def main():
    outdir, method = sys.argv[1], sys.argv[2]
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    data = make_dataset(n=240, p=384, ncausal=6, seed=77)
    X, y = data["X"], data["y"]
    lo, hi = shard_range(X.shape[1], rank, world, align=64)
    eng = OracleEngine("block")
    eng.load_dense(np.asfortranarray(X[:, lo:hi]))
    eng.setup_blocks(64)
    eng.init_state(method)
    if method == "BayesR":
        eng.set_state(delta=np.ones(hi - lo, dtype=np.int32))
    shard = MarkerShard(eng, lo, hi, rank, world)
    r = (y - y.mean()).astype(np.float32)[None, :]
    kw = (dict(vare=np.float32(0.5), var_effect=np.float32(0.05), pi_classes=np.array([0.95, 0.03, 0.015, 0.005]))
          if method == "BayesR" else dict(vare=np.float32(0.5), var_effect=np.float32(0.004), pi=0.9))
    stats = []
    for it in range(1, 6):
        r, st = shard.sweep(r, iteration=it, seed=5, **kw)
        stats.append([float(np.sum(st["sum_delta"])), float(st["alpha_ss"][0, 0]), float(st["resid_ss"][0, 0]),
                      float(st["n_events"]), float(np.sum(st["class_counts"]))])
    a, _, d = eng.get_state()
    np.savez(os.path.join(outdir, f"rank{rank}.npz"), r=r, alpha=a, delta=d, lo=lo, hi=hi, stats=np.array(stats))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
