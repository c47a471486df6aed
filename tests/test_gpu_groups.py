"""GROUPED LAUNCHES on the GPU (jwas_sweep_params.group_launch after jwas_hip_setup_groups; csrc/sweep.hpp k_group_step): every
launch of the step kernel streams 2 or 4 consecutive blocks and samples the previous group's blocks in order.

  * device vs the oracle's restatement of that schedule (oracle/jwas_oracle.c la_group_sweep): identical indicator / class
    trajectories, effects within a few float32 ulp -- the bar of every other device form;
  * device (grouped) vs the LITERAL non-block chain (BayesABC.jl:60-80, BayesR.jl:45-97): identical trajectories, effects
    <= 1e-4 of their scale (north_star's tolerance; the reference's own stream-vs-dense bar, test_streaming_codec.jl:100,104);
  * the flag without jwas_hip_setup_groups, and sweeps the mode does not cover, run the plain schedule bit for bit.
"""
import numpy as np
import pytest

from conftest import make_dataset
from oracle_engine import OracleEngine

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hip():
    import jwas_jl_amd as J
    e = J.HipEngine(0)
    yield e
    e.close()


def _hyper(data, pi=0.95):
    vare = np.float32(0.5 * data["y"].var())
    sum2pq = float((2 * data["freq"] * (1 - data["freq"])).sum())
    varg = np.float32(0.5 * data["y"].var() / ((1 - pi) * sum2pq))
    return vare, varg


def _engines(hip, data, bs, method, m, form="lookahead"):
    orc = OracleEngine(form=form)
    for e in (orc, hip):
        e.load_dense(data["X"])
        e.setup_blocks(bs, "f64")
        if form == "lookahead" or e is hip:
            e.setup_groups(m, "f64")
        e.init_state(method)
        e.set_residual(data["y"] - data["y"].mean())
    return orc, hip


def _kw(method, data, pi):
    vare, varg = _hyper(data, pi)
    if method == "BayesR":
        return dict(vare=vare, var_effect=np.float32(20 * varg), pi_classes=np.array([pi, 0.6 * (1 - pi), 0.3 * (1 - pi), 0.1 * (1 - pi)]))
    if method == "BayesB":
        rng = np.random.default_rng(5)
        return dict(vare=vare, var_effect=varg, pi=pi, var_effect_vec=(varg * rng.uniform(0.5, 2.0, data["X"].shape[1])).astype(np.float32))
    return dict(vare=vare, var_effect=varg, pi=pi)


def _same_state(orc, hip, atol=2e-6):
    ao, bo, do = orc.get_state(0)
    ah, bh, dh = hip.get_state(0)
    assert np.array_equal(do, dh), f"trajectories diverged at {np.flatnonzero(do != dh)[:5]}"
    np.testing.assert_allclose(ah, ao, rtol=0, atol=atol)
    np.testing.assert_allclose(hip.get_residual(0), orc.get_residual(0), rtol=0, atol=2e-5)


# p: a ragged last block and block counts that are and are not multiples of the group size (incl. a group of ONE block at the end,
# a sweep of a single group, and fewer blocks than a group holds)
@pytest.mark.parametrize("method,pi", [("BayesC", 0.9), ("BayesC", 0.5), ("BayesR", 0.9), ("BayesB", 0.8)])
@pytest.mark.parametrize("m", [2, 4])
@pytest.mark.parametrize("bs,nblk,tail", [(64, 9, 17), (64, 8, 0), (256, 5, 100), (256, 3, 0), (512, 4, 40), (1024, 3, 24), (64, 1, 0), (64, 0, 40)])
def test_grouped_chain_equals_its_oracle_restatement(hip, method, pi, m, bs, nblk, tail):
    if bs >= 512 and (method, pi) not in (("BayesC", 0.9), ("BayesR", 0.9)):
        pytest.skip("large blocks: the two headline methods")
    data = make_dataset(n=700, p=bs * nblk + tail, ncausal=10, seed=7 * bs + nblk + m)
    orc, hip = _engines(hip, data, bs, method, m)
    kw = _kw(method, data, pi)
    for it in range(1, 13):
        so = orc.sweep(iteration=it, seed=77, group_launch=True, **kw)
        sh = hip.sweep(iteration=it, seed=77, group_launch=True, **kw)
        assert so["n_events"] == sh["n_events"], f"iteration {it}"
    _same_state(orc, hip)


@pytest.mark.parametrize("method", ["BayesC", "BayesR"])
@pytest.mark.parametrize("m", [2, 4])
def test_grouped_chain_against_the_literal_oracle(hip, method, m):
    """The grouped schedule against the reference's own operation order (the non-block chain): same trajectories, effects within
    1e-4 of their scale; and the comparison is not vacuous (markers enter and leave the model)."""
    bs = 256
    data = make_dataset(n=5200, p=bs * 9 + 77, ncausal=25, seed=31 + m)
    orc, hip = _engines(hip, data, bs, method, m, form="dense")
    kw = _kw(method, data, 0.95)
    moved = 0
    for it in range(1, 16):
        so = orc.sweep(iteration=it, seed=9, **kw)
        sh = hip.sweep(iteration=it, seed=9, group_launch=True, **kw)
        moved += int(sh["n_events"])
    ao, _, do = orc.get_state(0)
    ah, _, dh = hip.get_state(0)
    assert np.array_equal(do, dh)
    assert moved > 200
    assert np.abs(ah - ao).max() <= 1e-4 * max(np.abs(ao).max(), 1e-3)
    rs = np.abs(orc.get_residual(0)).max()
    assert np.abs(hip.get_residual(0) - orc.get_residual(0)).max() <= 1e-4 * rs


def test_flag_is_ignored_where_the_mode_does_not_apply(hip):
    """group_launch without setup_groups, and on sweeps the mode does not cover (a uniform pi = 0, within-block repetitions): the plain
    schedule, bit for bit."""
    bs = 64
    data = make_dataset(n=600, p=bs * 6 + 11, ncausal=8, seed=3)
    res = {}
    for tag in ("plain", "flag_only", "groups_pi0"):
        hip.load_dense(data["X"])
        hip.setup_blocks(bs, "f64")
        if tag == "groups_pi0":
            hip.setup_groups(2, "f64")
        hip.init_state("BayesC")
        hip.set_residual(data["y"] - data["y"].mean())
        vare, varg = _hyper(data)
        for it in range(1, 6):
            pi = 0.0 if tag == "groups_pi0" or it == 5 else 0.9
            hip.sweep(iteration=it, seed=4, vare=vare, var_effect=varg, pi=pi, group_launch=(tag != "plain"),
                      nreps=(2 if it == 4 else 1))
        res[tag] = (hip.get_state(0), hip.get_residual(0))
    for a, b in zip(res["plain"][0], res["flag_only"][0]):
        assert np.array_equal(a, b)
    assert np.array_equal(res["plain"][1], res["flag_only"][1])
    # a chain under pi = 0 with the groups set up: every sweep takes the plain (dense) schedule
    hip.load_dense(data["X"])
    hip.setup_blocks(bs, "f64")
    hip.init_state("BayesC")
    hip.set_residual(data["y"] - data["y"].mean())
    vare, varg = _hyper(data)
    for it in range(1, 6):
        hip.sweep(iteration=it, seed=4, vare=vare, var_effect=varg, pi=0.0, nreps=(2 if it == 4 else 1))
    for a, b in zip(hip.get_state(0), res["groups_pi0"][0]):
        assert np.array_equal(a, b)


def test_groups_follow_the_selected_block_size(hip):
    """Groups belong to one resident block size: sweeps on the other size run the plain schedule; switching back and forth along a
    chain keeps the device on its oracle."""
    data = make_dataset(n=640, p=64 * 12 + 5, ncausal=9, seed=12)
    orc = OracleEngine(form="lookahead")
    for e in (orc, hip):
        e.load_dense(data["X"])
        e.setup_blocks(64, "f64")
        e.add_block_size(128, "f64")
        e.select_block_size(128)
        e.setup_groups(2, "f64")
        e.init_state("BayesC")
        e.set_residual(data["y"] - data["y"].mean())
    vare, varg = _hyper(data)
    for it in range(1, 11):
        bs = 128 if it % 3 else 64
        for e in (orc, hip):
            e.select_block_size(bs)
            e.sweep(iteration=it, seed=21, vare=vare, var_effect=varg, pi=0.85, group_launch=True)
    _same_state(orc, hip)


@pytest.mark.parametrize("seed", list(range(max(60, int(__import__("os").environ.get("JWAS_FUZZ_CASES", "400")) // 4))))
def test_random_grouped_launches_against_the_oracle(hip, seed):
    """Differential fuzzing of the grouped schedule: random shape (ragged n, ragged last block, fewer blocks than a group holds,
    several row groups), block size, group size, sampler, prior sparsity (very sparse ... half of the markers moving every sweep),
    per-marker priors, residual weights -- identical trajectories, effects within a few float32 ulp of the oracle's restatement."""
    rng = np.random.default_rng(50_000 + seed)
    method = str(rng.choice(["BayesC", "BayesC", "BayesR", "BayesB"]))
    m = int(rng.choice([2, 4]))
    bs = int(rng.choice([64, 64, 128, 256]))
    n = int(rng.integers(30, 700)) if rng.random() < 0.8 else int(rng.integers(1500, 4200))
    p = int(rng.integers(20, 14 * bs)) if bs <= 128 else int(rng.integers(200, 6 * bs))
    sp = float(rng.choice([0.3, 0.6, 0.9, 0.99]))
    d = make_dataset(n=n, p=p, ncausal=min(8, p), seed=seed)
    y = (d["y"] - d["y"].mean()).astype(np.float32)
    w = rng.uniform(0.3, 3.0, n).astype(np.float32) if rng.random() < 0.25 else None
    orc = OracleEngine("lookahead")
    for e in (orc, hip):
        e.load_dense(d["X"])
        e.set_weights(w)
        e.setup_blocks(bs, "f64")
        e.setup_groups(m, "f64")
        e.init_state(method)
        e.set_residual(y)
        if method == "BayesR":
            e.set_state(0, delta=np.ones(p, dtype=np.int32))
    v, g = np.float32(max(float(np.var(y)), 0.1)), np.float32(0.02)
    if method == "BayesC":
        kw = dict(vare=v, var_effect=g, pi=sp)
        if rng.random() < 0.25:
            kw = dict(vare=v, var_effect=g, pi_vec=np.clip(sp + rng.uniform(-0.2, 0.2, p), 0.01, 0.999))
    elif method == "BayesB":
        kw = dict(vare=v, var_effect=g, var_effect_vec=rng.uniform(0.005, 0.04, p).astype(np.float32), pi=sp)
    else:
        kw = dict(vare=v, var_effect=np.float32(0.1), pi_classes=np.concatenate([[sp], np.array([0.5, 0.3, 0.2]) * (1 - sp)]))
        if seed % 4 == 3:                                               # per-marker class priors (annotated BayesR, BayesR.jl:62-66)
            pm = np.random.default_rng(seed).dirichlet(np.ones(4), size=p) * 0.5 + 0.5 * kw["pi_classes"]
            kw["pi_matrix"] = pm / pm.sum(axis=1, keepdims=True)
    cfg = dict(method=method, m=m, bs=bs, n=n, p=p, sp=sp, weights=w is not None, marker_prior=("pi_matrix" in kw or "pi_vec" in kw))
    for it in range(1, 6):
        so = orc.sweep(iteration=it, seed=seed, group_launch=True, **kw)
        sh = hip.sweep(iteration=it, seed=seed, group_launch=True, **kw)
        assert so["n_events"] == sh["n_events"], f"{cfg} iteration {it}"
    ao, _, do = orc.get_state(0)
    ah, _, dh = hip.get_state(0)
    assert np.array_equal(do, dh), f"{cfg}"
    np.testing.assert_allclose(ah, ao, rtol=0, atol=1e-5, err_msg=str(cfg))
    np.testing.assert_allclose(hip.get_residual(0), orc.get_residual(0), rtol=0, atol=5e-5, err_msg=str(cfg))
    hip.set_weights(None)


@pytest.mark.parametrize("bs,m", [(64, 2), (64, 4), (256, 4), (512, 2), (1024, 4)])
def test_group_cross_grams_mfma_against_f64(hip, bs, m):
    """The pair / four cross-Grams from the MFMA kernel (k_cross_mfma128: fp32 products, fp32 chunks folded into fp64) against the
    exact fp64-accumulated ones (k_cross_f64): the same grouped chain -- every correction the chain takes from them agrees to
    float32 rounding, so trajectories and effects do (ragged last group, a last group of fewer blocks)."""
    data = make_dataset(n=900, p=bs * (2 * m + 1) + bs // 2 + 3, ncausal=12, seed=bs + m)
    kw = _kw("BayesC", data, 0.7)                       # (many changes per sweep: every cross-Gram block is exercised)
    res = {}
    for mode in ("f64", "mfma"):
        hip.load_dense(data["X"])
        hip.setup_blocks(bs, "f64")
        hip.setup_groups(m, mode)
        hip.init_state("BayesC")
        hip.set_residual(data["y"] - data["y"].mean())
        ev = 0
        for it in range(1, 9):
            ev += int(hip.sweep(iteration=it, seed=5, group_launch=True, **kw)["n_events"])
        res[mode] = (hip.get_state(0), hip.get_residual(0), ev)
    (a0, _, d0), r0, ev0 = res["f64"]
    (a1, _, d1), r1, _ = res["mfma"]
    assert ev0 > 8 * 0.1 * data["X"].shape[1]
    assert (d0 == d1).mean() >= 0.999
    same = d0 == d1
    assert np.abs(a0[same] - a1[same]).max() <= 1e-4 * max(np.abs(a0).max(), 1e-3)
    assert np.abs(r0 - r1).max() <= 2e-3 * np.abs(r0).max()


def test_setup_groups_contract_errors(hip):
    """jwas_hip_setup_groups: what it refuses (the error strings are the library's), and that 0 frees the groups again."""
    data = make_dataset(n=300, p=400, ncausal=4, seed=1)
    hip.load_dense(data["X"])
    hip.setup_blocks(64, "f64")
    with pytest.raises(RuntimeError, match="must be 0 .off., 2 or 4"):
        hip.setup_groups(3, "f64")
    hip.setup_groups(2, "f64")
    assert hip.blocks_per_launch() == 2
    hip.setup_groups(0, "f64")
    assert hip.blocks_per_launch() == 0
    hip.setup_blocks_explicit(np.array([0, 50, 130, 300]), "f64")
    with pytest.raises(RuntimeError, match="uniform blocks"):
        hip.setup_groups(2, "f64")
    hip.setup_blocks(64, "f64")                       # (leave the shared engine on a uniform partition)
