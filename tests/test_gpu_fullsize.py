"""BASELINE.json configs 3, 4 and 5 (one GPU's share) at their FULL workload sizes on the device.

The CPU oracle cannot run at these sizes (a 50 000 x 600 000 BayesR sweep takes it a quarter of an hour), so parity is
checked through the size-independent properties the domain offers, exactly as test_full_size_config2_invariants does:
  * residual identity        r = y - X alpha  after several sweeps (the sparse exit updates lose nothing),
  * statistics               every reduction the host draws consume equals a direct reduction of the returned state,
  * bit-reproducibility      the same seed gives the same chain, bit for bit,
  * block-size invariance    the draws do not depend on the block partition (only the fp32 rounding of the two Gram layouts
                             differs, so a vanishing fraction of indicators may flip),
  * the chain finds the simulated QTL.
The oracle comparison of the same kernels at oracle-sized inputs is tests/test_gpu_parity.py.
Reference shapes: BayesR.jl:45-97, MTBayesABC.jl:57-127, single_step/SSBR.jl:137-138.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

GAMMA = np.array([0.0, 0.01, 0.1, 1.0])


def _engine(need_bytes):
    import jwas_jl_amd as J
    e = J.HipEngine(0)
    if e.device_info()["hbm_free"] < need_bytes:
        e.close()
        pytest.skip(f"needs {need_bytes / 1e9:.0f} GB of free HBM")
    return e


def _phenotypes(e, n, p, t, nq, rng):
    """y_k = X a_k + e_k with nq QTL per trait (shared positions), h2 = 0.5; returns (Y t x n centred, idx, effects t x nq)."""
    idx = rng.choice(p, nq, replace=False)
    Y = np.empty((t, n), dtype=np.float32)
    eff = rng.standard_normal((t, nq))
    for k in range(t):
        a_true = np.zeros(p, dtype=np.float32)
        a_true[idx] = eff[k]
        e.set_state(0, alpha=a_true)
        g = e.mul_alpha(0).astype(np.float64)
        y = g / g.std() + rng.standard_normal(n)
        Y[k] = (y - y.mean()).astype(np.float32)
    return Y, idx, eff


def test_full_size_config3_bayesr_invariants():
    """Config 3's matrix on one GPU: single-trait BayesR, 50 000 x 600 000 fp32 (120 GB)."""
    n, p = 50_000, 600_000
    e = _engine(150e9)
    e.alloc_dense(n, p); e.synth(2026, 0, True)
    e.setup_blocks(512, "mfma"); e.add_block_size(1024, "mfma")
    e.init_state("BayesR")
    rng = np.random.default_rng(0)
    Y, idx, eff = _phenotypes(e, n, p, 1, 600, rng)
    y = Y[0]
    s2pq = float(e.xpx().astype(np.float64).sum()) / n
    pi0 = np.array([0.95, 0.03, 0.015, 0.005])
    sig = np.float32(1.0 / (s2pq * float((GAMMA * pi0).sum())))
    res = {}
    for tag, bs in (("a", 512), ("b", 1024), ("c", 512)):
        e.select_block_size(bs)
        e.set_state(alpha=np.zeros(p), delta=np.ones(p, dtype=np.int32))
        e.set_residual(y)
        pi = pi0.copy()
        for it in range(1, 6):
            st = e.sweep(iteration=it, seed=2026, vare=np.float32(1.0), var_effect=sig, pi_classes=pi)
            pi = (st["class_counts"] + 1.0) / (p + 4.0)
        a, _, dlt = e.get_state()
        r = e.get_residual()
        np.testing.assert_allclose(r, y - e.mul_alpha(), atol=5e-3)                              # residual identity
        assert np.array_equal(st["class_counts"], np.bincount(dlt, minlength=5)[1:5].astype(np.float64))
        assert ((dlt > 1) == (a != 0)).all()                                                     # class 1 <=> zero effect
        a64 = a.astype(np.float64)
        ssq = float((a64[dlt > 1] ** 2 / GAMMA[dlt[dlt > 1] - 1]).sum())                         # variance_components.jl:68-79
        assert st["bayesr_ssq"] == pytest.approx(ssq, rel=1e-9) and st["bayesr_nnz"] == float((dlt > 1).sum())
        assert st["resid_ss"][0, 0] == pytest.approx(float(r.astype(np.float64) @ r.astype(np.float64)), rel=1e-9)
        res[tag] = (a, dlt, r)
    e.close()
    assert np.array_equal(res["a"][0], res["c"][0]) and np.array_equal(res["a"][1], res["c"][1]) and np.array_equal(res["a"][2], res["c"][2])
    assert (res["a"][1] == res["b"][1]).mean() > 0.999                                           # block-size invariant draws
    both = (res["a"][1] == res["b"][1]) & (res["a"][1] > 1)
    assert np.abs(res["a"][0][both] - res["b"][0][both]).max() < 5e-3
    big = idx[np.abs(eff[0]) > 1.5]
    assert (res["a"][1][big] > 1).mean() > 0.5                                                   # the large QTL are in the model


@pytest.mark.parametrize("prior,bs", [("default", 128), ("sparse", 512)])
def test_full_size_config4_three_trait_invariants(prior, bs):
    """Config 4: 3-trait BayesC, Gibbs sampler I, 20 000 x 100 000 (8 GB); the reference's default prior (all mass on the
    all-ones state: every marker in the model, dense 128-marker blocks) and a sparse 8-state table (512-marker blocks)."""
    n, p, t = 20_000, 100_000, 3
    e = _engine(20e9)
    e.alloc_dense(n, p); e.synth(2026, 0, True)
    e.setup_blocks(bs, "mfma"); e.add_block_size(256, "mfma")
    e.init_state("MTBayesC", t)
    rng = np.random.default_rng(1)
    Y, idx, eff = _phenotypes(e, n, p, t, 100, rng)
    s2pq = float(e.xpx().astype(np.float64).sum()) / n
    ns = 1 << t
    pi0 = np.zeros(ns)
    if prior == "default":
        pi0[ns - 1] = 1.0
    else:
        pi0[0] = 0.95; pi0[1:] = 0.05 / (ns - 1)
    G = (np.eye(t) * 0.5 / (s2pq * pi0[ns - 1])).astype(np.float32)
    R = (np.eye(t) * 0.5 + 0.1).astype(np.float32)
    res = {}
    for tag, b in (("a", bs), ("b", 256), ("c", bs)):
        e.select_block_size(b)
        for k in range(t):
            e.set_state(k, alpha=np.zeros(p), beta=np.zeros(p), delta=np.ones(p))
            e.set_residual(Y[k], k)
        pi = pi0.copy()
        for it in range(1, 5):
            with np.errstate(divide="ignore"):
                st = e.sweep(iteration=it, seed=7, vare=R, var_effect=G, log_prior_states=np.log(pi))
            pi = (st["state_counts"] + 1.0) / (p + ns)
        A, B, D = (np.stack(v) for v in zip(*[e.get_state(k) for k in range(t)]))
        Rr = np.stack([e.get_residual(k) for k in range(t)])
        for k in range(t):
            np.testing.assert_allclose(Rr[k], Y[k] - e.mul_alpha(k), atol=5e-3)                  # residual identity per trait
        assert np.array_equal(A != 0, D != 0) and np.array_equal(A[D != 0], B[D != 0])           # alpha = delta * beta
        codes = (D != 0).T @ (1 << np.arange(t))
        assert np.array_equal(st["state_counts"], np.bincount(codes, minlength=ns).astype(np.float64))
        assert np.array_equal(st["sum_delta"], D.sum(axis=1).astype(np.float64))
        B64, R64 = B.astype(np.float64), Rr.astype(np.float64)
        np.testing.assert_allclose(st["beta_ss"], B64 @ B64.T, rtol=1e-9)                        # variance_components.jl:175-177
        np.testing.assert_allclose(st["resid_ss"], R64 @ R64.T, rtol=1e-9)
        res[tag] = (A, D, Rr)
    e.close()
    for q in range(3):
        assert np.array_equal(res["a"][q], res["c"][q])                                          # reproducible, bit for bit
    assert (res["a"][1] == res["b"][1]).mean() > 0.999                                           # block-size invariant draws
    if prior == "sparse":
        big = idx[np.abs(eff).max(axis=0) > 1.5]
        assert (res["a"][1][:, big] != 0).any(axis=0).mean() > 0.5


def test_full_size_config4_rule_t_against_the_sequential_chain():
    """Config 4 at full size under RULE T (jwas_sweep_params.section_solve: the dense 64-marker sections of the 256-marker blocks as
    triangular solves, MTBayesABC.jl:243-333 being the chain they stand for) against the SAME device's sequential chain on the same
    seeds: the same joint-state trajectory in every sweep, effects within 1e-4 of their scale (the reference's dense-vs-stream
    tolerance, test/unit/test_streaming_codec.jl:100,104), residual identity, statistics = direct reductions, bit-reproducibility;
    under the fixed all-ones prior EVERY section of every full block is solved without an exception; with pi estimated markers
    leave the model and are taken as exceptions inside the solved sections."""
    n, p, t = 20_000, 100_000, 3
    e = _engine(20e9)
    e.alloc_dense(n, p); e.synth(2026, 0, True)
    e.setup_blocks(256, "mfma")
    e.init_state("MTBayesC", t)
    rng = np.random.default_rng(1)
    Y, idx, eff = _phenotypes(e, n, p, t, 100, rng)
    s2pq = float(e.xpx().astype(np.float64).sum()) / n
    ns = 1 << t
    G = (np.eye(t) * 0.5 / s2pq).astype(np.float32)
    R = (np.eye(t) * 0.5 + 0.1).astype(np.float32)
    nsec = 4 * (p // 256)
    res = {}
    for tag, solve, est in (("solve", True, False), ("walk", False, False), ("again", True, False), ("solve_pi", True, True), ("walk_pi", False, True)):
        for k in range(t):
            e.set_state(k, alpha=np.zeros(p), beta=np.zeros(p), delta=np.ones(p))
            e.set_residual(Y[k], k)
        pi = np.zeros(ns); pi[ns - 1] = 1.0
        solved = fallen = exceptions = 0
        traj = []
        for it in range(1, 6):
            with np.errstate(divide="ignore"):
                st = e.sweep(iteration=it, seed=7, vare=R, var_effect=G, log_prior_states=np.log(pi), section_solve=solve)
            c = e.last_sweep_counters()
            solved += c[16]; fallen += c[17]; exceptions += c[23]
            traj.append(st["state_counts"].copy())
            if est:
                pi = (st["state_counts"] + 1.0) / (p + ns)
        A, B, D = (np.stack(v) for v in zip(*[e.get_state(k) for k in range(t)]))
        Rr = np.stack([e.get_residual(k) for k in range(t)])
        for k in range(t):
            np.testing.assert_allclose(Rr[k], Y[k] - e.mul_alpha(k), atol=5e-3)                  # residual identity per trait
        assert np.array_equal(A != 0, D != 0) and np.array_equal(A[D != 0], B[D != 0])           # alpha = delta * beta
        codes = (D != 0).T @ (1 << np.arange(t))
        assert np.array_equal(st["state_counts"], np.bincount(codes, minlength=ns).astype(np.float64))
        B64, R64 = B.astype(np.float64), Rr.astype(np.float64)
        np.testing.assert_allclose(st["beta_ss"], B64 @ B64.T, rtol=1e-9)
        np.testing.assert_allclose(st["resid_ss"], R64 @ R64.T, rtol=1e-9)
        if solve and not est:
            assert (solved, fallen, exceptions) == (5 * nsec, 0, 0)                              # nothing leaves the model: every section solved
        if solve and est:
            assert solved + fallen == 5 * nsec and solved > 0.9 * 5 * nsec and exceptions > 0    # markers do leave: exceptions at full size
        if not solve:
            assert solved == 0 and fallen == 0
        res[tag] = (A, D, Rr, traj)
    e.close()
    for q in range(3):
        assert np.array_equal(res["solve"][q], res["again"][q])                                  # reproducible, bit for bit
    # fixed all-ones prior: no marker can leave the model, so the two chains differ by the rule's float rounding only
    scale = float(np.abs(res["walk"][0]).max())
    assert np.array_equal(res["solve"][1], res["walk"][1])
    assert np.abs(res["solve"][0] - res["walk"][0]).max() < 1e-4 * scale
    assert (res["solve"][0] != res["walk"][0]).any()                                             # the rule is not a no-op
    # pi estimated: ~10 % of the sections see a marker leave and are walked; 1.5 million inclusion decisions whose right-hand sides
    # differ in the last bits may flip a handful of them (an MCMC chain is chaotic), everything else agrees to rounding
    for x, y in zip(res["solve_pi"][3], res["walk_pi"][3]):
        assert np.abs(x - y).max() <= 1e-3 * p
    same = res["solve_pi"][1] == res["walk_pi"][1]
    assert same.mean() > 0.999
    scale = float(np.abs(res["walk_pi"][0]).max())
    assert np.quantile(np.abs(res["solve_pi"][0] - res["walk_pi"][0])[same], 0.999) < 1e-4 * scale


def test_full_size_config5_shard_invariants():
    """One GPU's share of config 5 (single-step shaped input, SSBR.jl:137-138): 280 000 rows -- 80 000 integer-coded
    genotyped rows + 200 000 real-valued imputed rows -- x 75 000 markers fp32 (84 GB); single-trait BayesC."""
    n, p, n_gen = 280_000, 75_000, 80_000
    e = _engine(100e9)
    e.alloc_dense(n, p); e.synth_single_step(2026, n_gen, True)
    cols = e.get_columns(0, 2)
    raw = cols + (-cols[:n_gen]).max(axis=0)             # undo the centring: the genotyped rows hold codes 0/1/2 ...
    assert np.allclose(raw[:n_gen], np.round(raw[:n_gen]), atol=1e-4) and set(np.unique(np.round(raw[:n_gen]))) <= {0.0, 1.0, 2.0}
    frac = np.abs(raw[n_gen:] * 2 - np.round(raw[n_gen:] * 2)).max()
    assert frac < 1e-4 and (np.abs(raw[n_gen:] - np.round(raw[n_gen:])) > 0.4).any()             # ... the imputed rows halves
    e.setup_blocks(512, "mfma"); e.add_block_size(1024, "mfma")
    e.init_state("BayesC")
    rng = np.random.default_rng(2)
    Y, idx, eff = _phenotypes(e, n, p, 1, 75, rng)
    y = Y[0]
    xpx = e.xpx()
    assert xpx.min() > 0 and np.isfinite(xpx).all()
    varg = np.float32(1.0 / (0.05 * float(xpx.astype(np.float64).sum()) / n))
    res = {}
    for tag, bs in (("a", 512), ("b", 1024), ("c", 512)):
        e.select_block_size(bs)
        e.set_state(alpha=np.zeros(p), beta=np.zeros(p), delta=np.ones(p))
        e.set_residual(y)
        pi = 0.95
        for it in range(1, 7):
            st = e.sweep(iteration=it, seed=2026, vare=np.float32(1.0), var_effect=varg, pi=pi)
            pi = float(1 - (st["sum_delta"][0] + 1) / (p + 2))
        a, b, dlt = e.get_state()
        r = e.get_residual()
        np.testing.assert_allclose(r, y - e.mul_alpha(), atol=5e-3)                              # residual identity
        assert st["sum_delta"][0] == float(dlt.sum()) == float((a != 0).sum())
        assert st["alpha_ss"][0, 0] == pytest.approx(float(a.astype(np.float64) @ a.astype(np.float64)), rel=1e-9)
        assert st["resid_ss"][0, 0] == pytest.approx(float(r.astype(np.float64) @ r.astype(np.float64)), rel=1e-9)
        res[tag] = (a, dlt, r)
    e.close()
    assert np.array_equal(res["a"][0], res["c"][0]) and np.array_equal(res["a"][2], res["c"][2])    # reproducible, bit for bit
    assert (res["a"][1] == res["b"][1]).mean() > 0.9995                                         # block-size invariant draws
    big = idx[np.abs(eff[0]) > 1.5]
    assert (res["a"][1][big] != 0).mean() > 0.5
