"""N > 1 path on CPU: two gloo processes, each owning a marker shard (oracle engine), one all-reduce of
the residual delta per sweep.  The result must equal a single-process emulation of the reference's
independent-block semantics (BayesABC.jl:190-255) with one block per rank."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

import oracle as O
from conftest import make_dataset
from jwas_jl_amd.dist import shard_range

HERE = os.path.dirname(os.path.abspath(__file__))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def test_shard_range_is_aligned_and_covers():
    for p, w, al in ((600000, 8, 256), (384, 2, 64), (1000, 3, 64), (50, 4, 64)):
        rs = [shard_range(p, r, w, al) for r in range(w)]
        assert rs[0][0] == 0 and rs[-1][1] == p
        for (a, b), (c, d) in zip(rs[:-1], rs[1:]):
            assert b == c and a <= b
        assert all(lo % al == 0 for lo, _ in rs)


@pytest.mark.parametrize("method", ["BayesC", "BayesR"])
def test_two_rank_gloo_matches_independent_block_semantics(tmp_path, method):
    port = _free_port()
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), WORLD_SIZE="2")
    procs = [subprocess.Popen([sys.executable, os.path.join(HERE, "_dist_worker.py"), str(tmp_path), method],
                              env=dict(env, RANK=str(r)), stdout=subprocess.PIPE, stderr=subprocess.STDOUT) for r in range(2)]
    for pr in procs:
        out, _ = pr.communicate(timeout=300)
        assert pr.returncode == 0, out.decode()[-2000:]
    res = [np.load(tmp_path / f"rank{r}.npz") for r in range(2)]
    # every rank holds the same reconciled residual and the same global statistics
    assert np.array_equal(res[0]["r"], res[1]["r"])
    assert np.array_equal(res[0]["stats"], res[1]["stats"])

    # single-process emulation: both shards sweep from the same snapshot, then reconcile
    data = make_dataset(n=240, p=384, ncausal=6, seed=77)
    X, y = data["X"], data["y"]
    p = X.shape[1]
    shards = [shard_range(p, r, 2, 64) for r in range(2)]
    r = (y - y.mean()).astype(np.float32)
    alpha = np.zeros(p, dtype=np.float32)
    beta = np.zeros(p, dtype=np.float32)
    delta = np.ones(p, dtype=np.int32) if method == "BayesR" else np.zeros(p, dtype=np.float32)
    for it in range(1, 6):
        snap = r.copy()
        tot = np.zeros_like(r)
        for lo, hi in shards:
            Xs = np.asfortranarray(X[:, lo:hi])
            xpx = O.xpx(Xs)
            bs = O.block_starts_for(hi - lo, 64)
            gr = O.grams_for(Xs, bs)
            rl = snap.copy()
            a, b, d = alpha[lo:hi].copy(), beta[lo:hi].copy(), delta[lo:hi].copy()
            if method == "BayesR":
                O.bayesr_sweep(Xs, xpx, rl, a, d, 0.5, 0.05, [0.95, 0.03, 0.015, 0.005], 5, it, marker0=lo, block_starts=bs, grams=gr, nreps=1)
            else:
                O.bayesabc_sweep(Xs, xpx, rl, a, b, d, 0.5, 0.004, 0.9, 5, it, marker0=lo, block_starts=bs, grams=gr, nreps=1)
            alpha[lo:hi], beta[lo:hi], delta[lo:hi] = a, b, d
            tot += rl - snap
        r = snap + tot
    for k, (lo, hi) in enumerate(shards):
        assert (int(res[k]["lo"]), int(res[k]["hi"])) == (lo, hi)
        assert np.array_equal(res[k]["delta"], delta[lo:hi])
        np.testing.assert_allclose(res[k]["alpha"], alpha[lo:hi], atol=1e-7)
    np.testing.assert_allclose(res[0]["r"][0], r, atol=1e-5)
    assert res[0]["stats"][-1][2] == pytest.approx(float(r.astype(np.float64) @ r.astype(np.float64)), rel=1e-5)
