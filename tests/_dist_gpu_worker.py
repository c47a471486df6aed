"""Worker of tests/test_gpu_e2e.py::test_two_rank_sharded_sweep_over_rccl (launched with torch.distributed.run, 2 ranks)."""
import os
import sys

import numpy as np

sys.path[:0] = [os.environ["REPO"], os.path.join(os.environ["REPO"], "tests")]
import torch
import torch.distributed as dist
from conftest import make_dataset
import jwas_jl_amd as J
from jwas_jl_amd.dist import MarkerShard, shard_range

rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
torch.cuda.set_device(rank)
dist.init_process_group("nccl", rank=rank, world_size=world)
d = make_dataset(n=900, p=1536, ncausal=8, seed=21)
r0 = (d["y"] - d["y"].mean()).astype(np.float32)
kw = dict(seed=5, vare=np.float32(0.5), var_effect=np.float32(0.004), pi=0.9)
lo, hi = shard_range(1536, rank, world, align=256)
e = J.HipEngine(rank)
e.load_dense(np.asfortranarray(d["X"][:, lo:hi])); e.setup_blocks(256, "f64"); e.init_state("BayesC")
sh = MarkerShard(e, lo, hi, rank, world)
r = r0[None, :].copy()
for it in range(1, 5):
    r, st = sh.sweep(r, iteration=it, **kw)
a_mine = e.get_state()[0]
e.close()
if rank == 0:
    # emulation: both shards on this GPU, reconcile in fp64 on the host
    engs = []
    for g in range(world):
        l, h = shard_range(1536, g, world, align=256)
        eg = J.HipEngine(0)
        eg.load_dense(np.asfortranarray(d["X"][:, l:h])); eg.setup_blocks(256, "f64"); eg.init_state("BayesC")
        engs.append((eg, l, h))
    re = r0.copy()
    for it in range(1, 5):
        tot = np.zeros(len(re))
        nev = 0.0
        for eg, l, h in engs:
            eg.set_residual(re)
            s1 = eg.sweep(iteration=it, marker_offset=l, **kw)
            tot += eg.get_residual().astype(np.float64) - re.astype(np.float64)
            nev += s1["n_events"]
        re = (re.astype(np.float64) + tot).astype(np.float32)
    assert np.array_equal(re, r[0]), np.abs(re - r[0]).max()
    assert np.array_equal(engs[0][0].get_state()[0], a_mine)
    assert nev == st["n_events"]
    for eg, _, _ in engs:
        eg.close()
    print("TWO_RANK_RCCL_OK")
dist.barrier()
dist.destroy_process_group()
