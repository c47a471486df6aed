"""The oracle's GROUPED lookahead restatement (oracle/jwas_oracle.c la_group_sweep: the schedule of the device's grouped launches,
jwas_sweep_params.group_launch) pinned on the CPU to the forms it restates: the plain block chain (BayesABC.jl:145-187,
BayesR.jl:111-193) and the literal non-block chain (BayesABC.jl:60-80, BayesR.jl:45-97) -- identical indicator / class
trajectories, effects within float32 rounding -- and to the one-block lookahead when a group holds one block."""
import numpy as np
import pytest

from conftest import make_dataset
from oracle_engine import OracleEngine


def _hyper(data, pi):
    vare = np.float32(0.5 * data["y"].var())
    sum2pq = float((2 * data["freq"] * (1 - data["freq"])).sum())
    return vare, np.float32(0.5 * data["y"].var() / ((1 - pi) * sum2pq))


def _run(data, form, method, bs, m, nit, kw):
    e = OracleEngine(form=form)
    e.load_dense(data["X"])
    e.setup_blocks(bs)
    if m:
        e.setup_groups(m)
    e.init_state(method)
    e.set_residual(data["y"] - data["y"].mean())
    moved = 0
    for it in range(1, nit + 1):
        moved += int(e.sweep(iteration=it, seed=11, group_launch=bool(m), **kw)["n_events"])
    return e.get_state(0), e.get_residual(0), moved


@pytest.mark.parametrize("method", ["BayesC", "BayesR"])
@pytest.mark.parametrize("m", [2, 4])
@pytest.mark.parametrize("bs,p", [(32, 32 * 7 + 9), (64, 64 * 4), (64, 50)])
def test_grouped_lookahead_is_the_block_chain(method, m, bs, p):
    data = make_dataset(n=300, p=p, ncausal=6, seed=bs + m)
    vare, varg = _hyper(data, 0.8)
    kw = (dict(vare=vare, var_effect=np.float32(20 * varg), pi_classes=np.array([0.8, 0.12, 0.06, 0.02])) if method == "BayesR"
          else dict(vare=vare, var_effect=varg, pi=0.8))
    (ag, _, dg), rg, moved = _run(data, "lookahead", method, bs, m, 12, kw)
    for form in ("block", "dense"):
        (ab, _, db), rb, _ = _run(data, form, method, bs, 0, 12, kw)
        assert np.array_equal(dg, db), form
        assert np.abs(ag - ab).max() <= 2e-5 * max(np.abs(ab).max(), 1e-3), form
        assert np.abs(rg - rb).max() <= 1e-4 * np.abs(rb).max(), form
    assert moved > 20                                   # (not vacuous: markers entered and left the model)


def test_group_of_one_block_is_the_one_block_lookahead():
    """A partition with fewer blocks than a group holds, and the flag without setup_groups: the one-block lookahead's bits where
    the two schedules coincide (a single block: no correction at all)."""
    data = make_dataset(n=200, p=40, ncausal=4, seed=2)
    vare, varg = _hyper(data, 0.7)
    kw = dict(vare=vare, var_effect=varg, pi=0.7)
    (a1, b1, d1), r1, _ = _run(data, "lookahead", "BayesC", 64, 0, 6, kw)
    (a2, b2, d2), r2, _ = _run(data, "lookahead", "BayesC", 64, 2, 6, kw)
    assert np.array_equal(a1, a2) and np.array_equal(d1, d2) and np.array_equal(r1, r2)
