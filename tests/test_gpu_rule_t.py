"""RULE T on the GPU (jwas_sweep_params.section_solve; csrc/sampler_mt.hpp, sampler_st.hpp): dense 64-marker sections of full blocks
evaluated as a triangular solve with the section's per-sweep inverse instead of a 64-step walk.

Reference chains it stands for: MTBayesABC.jl:243-333 (block form of sampler I), BayesABC.jl:153-185 (single-trait block form under
Pi = 0: benchmarks/jwas_nonblock_benchmark.jl:34-51).  Three kinds of test:
  * device vs the oracle's own restatement of the rule (orc mt1_section_solve / abc_section_solve): identical indicator trajectories,
    effects within 5e-6, every sweep's change count equal -- and, with the oracle's sums in the device's order, bit for bit;
  * device vs the LITERAL oracle (no Rule L / D / T: the reference's operation order): identical inclusion trajectories, effects
    within 1e-4 of their scale (the tolerance the reference accepts between its own dense and streaming paths,
    test/unit/test_streaming_codec.jl:100,104) at the production geometries (t = 3 / 256-marker blocks; 512-marker blocks; n >= 5000);
  * the solve really ran (jwas_hip_last_sweep_counters), and a prior under which verification fails falls back to the walk."""
import numpy as np
import pytest

from conftest import make_dataset
from oracle_engine import OracleEngine
import oracle as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hip():
    import jwas_jl_amd as J
    e = J.HipEngine(0)
    yield e
    e.close()


def _mt_setup(hip, data, t, method, rng, gram_mode="f64"):
    orc = OracleEngine(form="lookahead")
    for e in (orc, hip):
        e.load_dense(data["X"])
        if e is hip:
            e.setup_blocks(256, gram_mode)
        else:
            e.setup_blocks(256)
        e.init_state(method, t)
    Y = np.stack([(1 + 0.2 * k) * (data["y"] - data["y"].mean()) + 0.3 * rng.standard_normal(len(data["y"])).astype(np.float32)
                  for k in range(t)]).astype(np.float32)
    for k in range(t):
        for e in (orc, hip):
            e.set_residual(Y[k], k)
            e.set_state(k, delta=np.ones(e.p, dtype=np.float32))
    return orc, hip


def _mt_hyper(t, rng):
    A = rng.standard_normal((t, t)); B = rng.standard_normal((t, t))
    vare = ((A @ A.T / t + np.eye(t)) * 0.5).astype(np.float32)
    varg = ((B @ B.T / t + np.eye(t)) * 0.003).astype(np.float32)
    return vare, varg


@pytest.mark.parametrize("method,t,leak", [("MTBayesC", 3, 1e-9), ("MTBayesC", 2, 1e-9), ("MTBayesB", 3, 1e-9), ("MTBayesC", 3, 2e-3),
                                           ("MTBayesC", 3, 6e-2), ("MTBayesB", 2, 2e-2)])
def test_rule_t_multitrait_device_vs_its_oracle_restatement(hip, method, t, leak):
    """Sampler I, 256-marker blocks (three full ones and a ragged tail), every marker in the model at the start.  leak = the prior
    mass of every other joint state: at 2e-3 markers do leave the model -- the solve takes them as EXCEPTIONS (their literal
    evaluation replaces their row of the solution, the rows behind take a rank-t correction), also the markers that are outside
    the model when a later sweep enters their section; at 6e-2 sections collect more exceptions than the rule allows and fall
    back to the walk, or are not tried at all -- the device and the oracle must take the same decisions everywhere."""
    rng = np.random.default_rng(70 + t)
    data = make_dataset(n=1100, p=3 * 256 + 77, ncausal=14, seed=700 + t)
    orc, hip = _mt_setup(hip, data, t, method, rng)
    vare, varg = _mt_hyper(t, rng)
    prior = np.full(1 << t, leak); prior[-1] = 1.0; prior /= prior.sum()
    kw = dict(vare=vare, var_effect=varg, log_prior_states=np.log(prior), section_solve=True)
    if method == "MTBayesB":
        Vm = np.stack([varg * rng.uniform(0.6, 1.6) for _ in range(orc.p)]).astype(np.float32)
        kw["var_effect_matrix"] = Vm
    solved = fallen = exceptions = 0
    O.section_solve_counts(reset=True)
    for it in range(1, 9):
        so = orc.sweep(iteration=it, seed=31, **kw)
        sh = hip.sweep(iteration=it, seed=31, **kw)
        cnt = hip.last_sweep_counters()
        solved += cnt[16]; fallen += cnt[17]; exceptions += cnt[23]
        assert so["n_events"] == sh["n_events"], f"iteration {it}"
        assert np.array_equal(so["state_counts"], sh["state_counts"]), f"iteration {it}"
        for k in range(t):
            ao, bo, do = orc.get_state(k)
            ah, bh, dh = hip.get_state(k)
            assert np.array_equal(do, dh), f"iteration {it}, trait {k}: indicators differ at {np.flatnonzero(do != dh)[:5]}"
            np.testing.assert_allclose(ah, ao, rtol=0, atol=5e-6)
            np.testing.assert_allclose(bh, bo, rtol=0, atol=5e-6)
    o_solved, o_fallen = O.section_solve_counts()
    assert (solved, fallen) == (o_solved, o_fallen)          # the same sections solved / fallen back on both sides
    assert exceptions == O.section_solve_exceptions()       # ... and the same number of exceptions taken inside the solved ones
    assert solved > 0
    if leak > 1e-3:
        assert exceptions > 0                                # the exception path was exercised
    if leak >= 5e-2:
        assert fallen > 0                                    # ... and the fallback
    for k in range(t):
        np.testing.assert_allclose(hip.get_residual(k), orc.get_residual(k), rtol=0, atol=3e-5)


def test_rule_t_off_is_the_sequential_chain_bit_for_bit(hip):
    """section_solve = 0 (the default): nothing changes -- the chain of dense_big_mt, bit for bit, on the same engine that just ran
    with the rule on; and the two chains differ (the rule is not a no-op) while agreeing to rounding."""
    t = 3
    rng = np.random.default_rng(5)
    data = make_dataset(n=900, p=4 * 256, ncausal=12, seed=63)
    vare, varg = _mt_hyper(t, rng)
    prior = np.full(1 << t, 1e-9); prior[-1] = 1.0; prior /= prior.sum()
    out = {}
    for tag, solve in (("solve", True), ("walk", False), ("oracle_walk", False)):
        if tag == "oracle_walk":
            e = OracleEngine("lookahead"); e.load_dense(data["X"]); e.setup_blocks(256)
        else:
            e = hip; e.load_dense(data["X"]); e.setup_blocks(256, "f64")
        e.init_state("MTBayesC", t)
        y = data["y"] - data["y"].mean()
        for k in range(t):
            e.set_residual(((1 + 0.25 * k) * y).astype(np.float32), k)
            e.set_state(k, delta=np.ones(e.p, dtype=np.float32))
        for it in range(1, 8):
            e.sweep(iteration=it, seed=23, vare=vare, var_effect=varg, log_prior_states=np.log(prior), section_solve=solve)
        out[tag] = [e.get_state(k) for k in range(t)]
        if tag == "walk":
            assert hip.last_sweep_counters()[16] == 0
    differ = 0
    for k in range(t):
        np.testing.assert_allclose(out["walk"][k][0], out["oracle_walk"][k][0], rtol=0, atol=5e-6)
        assert np.array_equal(out["walk"][k][2], out["oracle_walk"][k][2])
        scale = float(np.abs(out["walk"][k][0]).max())
        np.testing.assert_allclose(out["solve"][k][0], out["walk"][k][0], rtol=0, atol=1e-5 * scale)
        differ += int((out["solve"][k][0] != out["walk"][k][0]).sum())
    assert differ > 0


@pytest.mark.parametrize("method", ["MTBayesC", "MTBayesB"])
def test_rule_t_device_against_the_literal_oracle_t3_bs256(hip, method):
    """VERDICT r04 item 1's parity bar: the device under Rule T against the oracle in the reference's LITERAL order
    (_MTBayesABC_samplerI!, MTBayesABC.jl:85-120: no linear form, no solve) at config 4's geometry -- three traits, 256-marker
    blocks, the reference's default all-ones prior, n = 5 200 -- identical inclusion trajectories in every sweep, effects within
    1e-4 of their scale after twelve sweeps; production MFMA Grams on the device."""
    t = 3
    rng = np.random.default_rng(91)
    data = make_dataset(n=5200, p=5 * 256, ncausal=25, seed=4300)
    vare, varg = _mt_hyper(t, rng)
    prior = np.zeros(1 << t); prior[-1] = 1.0                       # tools4genotypes.jl:357-373: all mass on the all-ones state
    with np.errstate(divide="ignore"):
        lp = np.log(prior)
    kw = dict(vare=vare, var_effect=varg, log_prior_states=lp)
    try:
        O.lib().orc_set_mt_linear_form(0)
        orc, hip = _mt_setup(hip, data, t, method, rng, gram_mode="mfma")
        if method == "MTBayesB":
            kw["var_effect_matrix"] = np.stack([varg * rng.uniform(0.6, 1.6) for _ in range(orc.p)]).astype(np.float32)
        solved = 0
        for it in range(1, 13):
            so = orc.sweep(iteration=it, seed=37, **kw)                              # literal: no section_solve
            sh = hip.sweep(iteration=it, seed=37, section_solve=True, **kw)
            solved += hip.last_sweep_counters()[16]
            assert np.array_equal(so["state_counts"], sh["state_counts"]), f"iteration {it}"
            for k in range(t):
                assert np.array_equal(orc.get_state(k)[2], hip.get_state(k)[2]), f"iteration {it}, trait {k}"
    finally:
        O.lib().orc_set_mt_linear_form(1)
    assert solved == 12 * 5 * 4                                     # every section of every block of every sweep was solved
    for k in range(t):
        ao, ah = orc.get_state(k)[0], hip.get_state(k)[0]
        scale = max(float(np.abs(ao).max()), 1e-3)
        np.testing.assert_allclose(ah, ao, rtol=0, atol=1e-4 * scale)


@pytest.mark.timeout(120)
def test_four_traits_ignore_the_flag(hip):
    """jwas_hip.h: '<= 3 traits ... every other sweep ignores the flag'.  With 4 traits the sampler has no solve path (the
    host used to set up the hand-over words anyway and the helper workgroup waited for sections nobody published): section_solve = 1
    must run the sequential walk, bit for bit."""
    t = 4
    rng = np.random.default_rng(9)
    data = make_dataset(n=800, p=3 * 256 + 40, ncausal=10, seed=81)
    vare, varg = _mt_hyper(t, rng)
    prior = np.full(1 << t, 1e-9); prior[-1] = 1.0; prior /= prior.sum()
    out = {}
    for solve in (True, False):
        hip.load_dense(data["X"]); hip.setup_blocks(256, "f64")
        hip.init_state("MTBayesC", t)
        y = data["y"] - data["y"].mean()
        for k in range(t):
            hip.set_residual(((1 + 0.25 * k) * y).astype(np.float32), k)
            hip.set_state(k, delta=np.ones(hip.p, dtype=np.float32))
        for it in range(1, 5):
            hip.sweep(iteration=it, seed=23, vare=vare, var_effect=varg, log_prior_states=np.log(prior), section_solve=solve)
        assert hip.last_sweep_counters()[16] == 0
        out[solve] = [hip.get_state(k) for k in range(t)] + [hip.get_residual(k) for k in range(t)]
    for a, b in zip(out[True], out[False]):
        assert np.array_equal(np.asarray(a), np.asarray(b))
