"""The device schedules AT THE GEOMETRIES bench.py TIMES against the LITERAL per-marker chain (oracle form 'dense': one dot product,
one scalar update, one axpy per marker -- BayesABC.jl:60-80, BayesR.jl:45-97), not against a device-shaped restatement.

Rule: no schedule ships without a literal-chain test at the geometry bench.py times.  NOTES R5 records why: the
first grouped version double-counted a correction, device and restatement agreed with each other across 58 green cases, and only
the literal test caught it.  Cases here:

  * grouped launches of 1024-marker blocks, m = 2 and 4 blocks per launch, MFMA Grams and MFMA group cross-Grams (the headline:
    bench.py config2), p >= 13 000 so that three full groups + a ragged one run and cG (group -> group), cP (pair -> pair) and cW
    (block -> block) all form; n = 5 200 (21 row slices, several row groups);
  * the same from 2-bit packed storage (bench.py --storage packed2bit): literal chain on the DECODED matrix;
  * 512-marker pairs (2 blocks per launch, the ping-pong samplers of the sampler-bound sweeps: bench.py config3, --pi-fixed);
  * one block per launch at 512 markers with MANY candidates per block -- the compact candidate chain of config 3 (BayesR) and of
    a fixed pi (BayesC) -- and at 1024 markers (the compact chain from one candidate on).

Tolerance: identical indicator / class trajectories, effects and residual within 1e-4 of their scale (north_star's stated
floating-point tolerance; the reference's own stream-vs-dense bar, test/unit/test_streaming_codec.jl:100,104).
"""
import numpy as np
import pytest

import oracle as O
from conftest import make_dataset
from oracle_engine import OracleEngine
from jwas_jl_amd import streaming as S

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hip():
    import jwas_jl_amd as J
    e = J.HipEngine(0)
    yield e
    e.close()


def _literal(X, method, y):
    """The literal non-block oracle on X (no Grams are formed: the dense form only needs x'x)."""
    orc = OracleEngine(form="dense")
    orc.load_dense(X)
    orc._xpx = O.xpx(orc.X, orc.acc)
    orc._groups = {}
    orc.block_size = 0
    orc.init_state(method)
    orc.set_residual(y)
    return orc


def _kw(method, y, freq, pi):
    vare = np.float32(0.5 * y.var())
    sum2pq = float((2 * freq * (1 - freq)).sum())
    varg = np.float32(0.5 * y.var() / ((1 - pi) * sum2pq))
    if method == "BayesR":
        return dict(vare=vare, var_effect=np.float32(20 * varg), pi_classes=np.array([pi, 0.6 * (1 - pi), 0.3 * (1 - pi), 0.1 * (1 - pi)]))
    return dict(vare=vare, var_effect=varg, pi=pi)


def _compare(orc, hip, moved, min_moved):
    ao, _, do = orc.get_state(0)
    ah, _, dh = hip.get_state(0)
    assert np.array_equal(do, dh), f"trajectories diverged at markers {np.flatnonzero(do != dh)[:8]}"
    assert moved > min_moved, f"vacuous comparison: only {moved} effect changes"
    assert np.abs(ah - ao).max() <= 1e-4 * max(np.abs(ao).max(), 1e-3)
    rs = np.abs(orc.get_residual(0)).max()
    assert np.abs(hip.get_residual(0) - orc.get_residual(0)).max() <= 1e-4 * rs


@pytest.mark.parametrize("method", ["BayesC", "BayesR"])
@pytest.mark.parametrize("m,bs", [(2, 1024), (4, 1024), (2, 512), (4, 512)])
def test_grouped_1024_marker_launches_against_the_literal_chain(hip, method, m, bs):
    """bench.py's headline schedule (k_group_step, 1024-marker blocks, MFMA Grams + k_cross_mfma128 group cross-Grams): 13 full
    blocks + a ragged one = three full groups of four and a group of two (m = 4) / seven pairs (m = 2); and the 512-marker PAIRS / FOURS of
    the high-turnover sweeps (config 3 / a fixed pi: the ping-pong samplers, one sampler workgroup per block -- the first sweep of a chain and every sweep in which more
    than 1.25 % of the markers changed run them)."""
    data = make_dataset(n=5200, p=1024 * 13 + 300, ncausal=40, seed=600 + m)
    y = (data["y"] - data["y"].mean()).astype(np.float32)
    orc = _literal(data["X"], method, y)
    hip.load_dense(data["X"])
    hip.setup_blocks(bs, "mfma")
    hip.setup_groups(m, "mfma")
    hip.init_state(method)
    hip.set_residual(y)
    kw = _kw(method, y, data["freq"], 0.97)
    moved = 0
    for it in range(1, 13):
        so = orc.sweep(iteration=it, seed=19, **kw)
        sh = hip.sweep(iteration=it, seed=19, group_launch=True, **kw)
        assert so["n_events"] == sh["n_events"], f"iteration {it}"
        moved += int(sh["n_events"])
    assert hip.blocks_per_launch() == m
    _compare(orc, hip, moved, 1000)


@pytest.mark.parametrize("method,m", [("BayesC", 4), ("BayesR", 2)])
def test_packed_grouped_1024_marker_launches_against_the_literal_chain(hip, method, m):
    """bench.py --storage packed2bit (k_group_step<., PackedCols>, update_role_wide): the literal chain on the decoded matrix.
    n = 5 200 = five full 1024-row slices and a ragged one; missing codes present; two full groups and a ragged one."""
    bs, n = 1024, 5200
    p = bs * (2 * m + 1) + 77
    d = make_dataset(n=n, p=p, ncausal=30, seed=77 + m, center=False)
    raw = d["raw"].astype(np.float64)
    rng = np.random.default_rng(5)
    raw[rng.integers(0, n, 300), rng.integers(0, p, 300)] = 9
    miss = raw == 9
    codes = np.where(miss, 3, raw).astype(np.uint8)
    means = np.array([raw[~miss[:, j], j].mean(dtype=np.float32) for j in range(p)], dtype=np.float32)
    v = np.where(miss, means[None, :], raw.astype(np.float32)).astype(np.float32)
    X = np.asfortranarray(v - means[None, :])
    y = (d["y"] - d["y"].mean()).astype(np.float32)
    orc = _literal(X, method, y)
    hip.load_packed2bit(S.pack_2bit(codes), n, means, centered=True)
    hip.setup_blocks(bs, "mfma")
    hip.setup_groups(m, "mfma")
    hip.init_state(method)
    hip.set_residual(y)
    kw = _kw(method, y, means / 2.0, 0.97)
    moved = 0
    for it in range(1, 11):
        so = orc.sweep(iteration=it, seed=23, **kw)
        sh = hip.sweep(iteration=it, seed=23, group_launch=True, **kw)
        assert so["n_events"] == sh["n_events"], f"iteration {it}"
        moved += int(sh["n_events"])
    _compare(orc, hip, moved, 500)


@pytest.mark.parametrize("method,bs,pi", [("BayesR", 512, 0.95), ("BayesC", 512, 0.95), ("BayesR", 1024, 0.99), ("BayesC", 1024, 0.995)])
def test_compact_candidate_chain_against_the_literal_chain(hip, method, bs, pi):
    """One block per launch with a sparse prior and many markers in the model (config 3 / --pi-fixed: 512-marker blocks, some tens of
    candidates each) and 1024-marker blocks with a handful: the COMPACT candidate chain (sampler_st.hpp compact_walk) must be what
    ran (sweep counter 16 = blocks that tried it) and must be the literal chain."""
    data = make_dataset(n=5200, p=bs * 9 + 131, ncausal=30, seed=900 + bs)
    y = (data["y"] - data["y"].mean()).astype(np.float32)
    orc = _literal(data["X"], method, y)
    hip.load_dense(data["X"])
    hip.setup_blocks(bs, "mfma")
    hip.init_state(method)
    hip.set_residual(y)
    kw = _kw(method, y, data["freq"], pi)
    moved, tried = 0, 0
    for it in range(1, 13):
        so = orc.sweep(iteration=it, seed=41, **kw)
        sh = hip.sweep(iteration=it, seed=41, **kw)
        assert so["n_events"] == sh["n_events"], f"iteration {it}"
        moved += int(sh["n_events"])
        c = hip.last_sweep_counters()
        tried += int(c[16]) - int(c[17])
    # (512-marker blocks: most blocks hold 6..64 candidates; 1024-marker blocks take the chain from one candidate on but hold at most
    # 32 staged rows, so the first sweeps of a chain -- many candidates per block -- run the speculative rounds there)
    assert tried >= (60 if bs == 512 else 12), f"the compact chain ran in only {tried} blocks"
    _compare(orc, hip, moved, 500)
