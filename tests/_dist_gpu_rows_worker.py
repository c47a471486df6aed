"""Worker of tests/test_gpu_e2e.py::test_two_rank_row_shards_over_rccl (launched with torch.distributed.run, 2 ranks):
exact row shards over RCCL against the same run over the loopback transport (two threads on GPU 0) -- the two transports
must give the same chain bit for bit (a two-term sum does not depend on the order)."""
import os
import sys
import threading

import numpy as np

sys.path[:0] = [os.environ["REPO"], os.path.join(os.environ["REPO"], "tests")]
import torch
import torch.distributed as dist
from conftest import make_dataset
import jwas_jl_amd as J

rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
torch.cuda.set_device(rank)
dist.init_process_group("nccl", rank=rank, world_size=world)
rows = 512
d = make_dataset(n=rows * world, p=700, ncausal=8, seed=33)
y = (d["y"] - d["y"].mean()).astype(np.float32)
kw = dict(seed=5, vare=np.float32(0.5), var_effect=np.float32(0.004), pi=0.9)


def chain(e, r, attach):
    sl = slice(r * rows, (r + 1) * rows)
    e.load_dense(np.asfortranarray(d["X"][sl]))
    attach(e, r)
    e.comm_row_shards(True)
    e.setup_blocks(128, "f64"); e.init_state("BayesC"); e.set_residual(y[sl])
    st = [e.sweep(iteration=it, **kw) for it in range(1, 6)]
    return e.get_state()[0], e.get_residual(), st[-1]["resid_ss"]


def attach_rccl(e, r):
    box = [e.comm_unique_id() if r == 0 else None]
    dist.broadcast_object_list(box, src=0)
    e.comm_init(box[0], r, world)


e = J.HipEngine(rank)
a_mine, r_mine, ss = chain(e, rank, attach_rccl)
e.close()
if rank == 0:
    out = [None] * world

    def run(r):
        eg = J.HipEngine(0)
        out[r] = chain(eg, r, lambda en, rr: en.comm_init_loopback(3, rr, world))
        eg.close()
    th = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    for x in th:
        x.start()
    for x in th:
        x.join(timeout=300)
    assert np.array_equal(out[0][0], a_mine) and np.array_equal(out[0][1], r_mine) and np.array_equal(out[0][2], ss)
    print("TWO_RANK_ROWS_RCCL_OK")
dist.barrier()
dist.destroy_process_group()
