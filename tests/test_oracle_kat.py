"""CPU tests that PIN THE ORACLE against everything the reference's own tests hold for this path
(SURVEY.md section 8c): closed-form / deterministic items only -- the reference ships no golden
vectors for sampler output, so output values on a seed stay "parity unpinned" (oracle header).
Each test cites the reference test it restates.
"""
import numpy as np
import pytest

import oracle as O
from conftest import make_dataset


# ---- RNG: Philox4x32-10 known-answer vectors (Random123 kat_vectors) -----------------------------
@pytest.mark.parametrize("ctr,key,exp", [
    ([0, 0, 0, 0], [0, 0], [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]),
    ([0xffffffff] * 4, [0xffffffff] * 2, [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]),
    ([0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344], [0xa4093822, 0x299f31d0],
     [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1]),
])
def test_philox_known_answers(ctr, key, exp):
    assert O.philox(ctr, key).tolist() == exp


def test_rng_moments_and_ranges():
    u = np.array([O.uniform(7, m, 1) for m in range(50000)])
    z = np.array([O.normal(7, m, 1) for m in range(50000)])
    assert 0.0 < u.min() and u.max() < 1.0
    assert abs(u.mean() - 0.5) < 0.01 and abs(u.var() * 12 - 1) < 0.03
    assert abs(z.mean()) < 0.02 and abs(z.std() - 1) < 0.02
    assert abs(np.corrcoef(u, z)[0, 1]) < 0.02
    # draws are keyed by (marker, iteration, rep, trait): changing any of them changes the draw
    base = O.uniform(1, 5, 3, 0, 0)
    assert len({base, O.uniform(2, 5, 3), O.uniform(1, 6, 3), O.uniform(1, 5, 4), O.uniform(1, 5, 3, 1), O.uniform(1, 5, 3, 0, 1)}) == 6


# ---- test/unit/test_bayesr.jl:244-250 ------------------------------------------------------------
def test_bayesr_block_nreps_schedule():
    assert O.bayesr_block_nreps(1, 10, 7) == 1
    assert O.bayesr_block_nreps(10, 10, 7) == 1
    assert O.bayesr_block_nreps(11, 10, 7) == 7
    assert O.bayesr_block_nreps(25, 0, 7) == 7
    assert O.bayesr_block_nreps(3, 8, 1) == 1
    with pytest.raises(ValueError):
        O.bayesr_block_nreps(1, 0, 0)


# ---- test/unit/test_bayesr.jl:252-262 ------------------------------------------------------------
def test_bayesr_sigma_sufficient_statistics():
    alpha = np.array([0.0, 0.4, -0.3, 0.1, 0.0])
    delta = np.array([1, 2, 4, 3, 1])
    gamma = np.array([0.0, 0.01, 0.1, 1.0])
    ssq, nnz = O.bayesr_sigma_suffstats(alpha, delta, gamma)
    a32 = alpha.astype(np.float32).astype(np.float64)
    expected = a32[1] ** 2 / gamma[1] + a32[2] ** 2 / gamma[3] + a32[3] ** 2 / gamma[2]
    assert ssq == pytest.approx(expected, rel=1e-12)
    assert nnz == 3
    # test_bayesr.jl:264-280: sigma2 = (ssq + df*scale) / chisq(nnz + df) is the host formula
    chi = 5.3
    assert (ssq + 4.0 * 0.2) / chi == pytest.approx((expected + 0.8) / chi)


# ---- test/unit/test_bayesr.jl:202-242: dense and block BayesR kernels on the 4x2 toy ---------------
def _toy():
    X = np.asfortranarray(np.array([[0, 2], [1, 1], [2, 0], [1, 1]], dtype=np.float32))
    return X, O.xpx(X), np.array([0.8, -0.1, 0.3, 0.5], dtype=np.float32)


@pytest.mark.parametrize("block", [False, True])
def test_bayesr_toy_classes_in_range(block):
    X, xpx, y = _toy()
    alpha, delta = np.zeros(2, dtype=np.float32), np.ones(2, dtype=np.int32)
    kw = dict(block_starts=np.array([0]), grams=O.gram(X, 0, 2).ravel(), nreps=0) if block else {}
    O.bayesr_sweep(X, xpx, y.copy(), alpha, delta, 1.0, 0.2, [0.95, 0.03, 0.015, 0.005], 1234, 1, **kw)
    assert np.all((1 <= delta) & (delta <= 4)) and alpha.shape == (2,)


# ---- test/unit/test_annotated_bayesr.jl:247-266: degenerate per-SNP priors, RNG independent -------
@pytest.mark.parametrize("seed", [1, 20260327, 99])
def test_bayesr_degenerate_priors_force_classes(seed):
    X, xpx, y = _toy()
    alpha, delta = np.zeros(2, dtype=np.float32), np.ones(2, dtype=np.int32)
    snp_pi = np.array([[0.0, 1.0, 0.0, 0.0], [1.0, 0.0, 0.0, 0.0]])
    O.bayesr_sweep(X, xpx, y.copy(), alpha, delta, 1.0, 0.2, snp_pi, seed, 1)
    assert delta.tolist() == [2, 1]
    assert alpha[0] != 0 and alpha[1] == 0


# ---- BayesR.jl:9-20,50 error contract ---------------------------------------------------------------
def test_bayesr_prior_validation_errors():
    X, xpx, y = _toy()
    a, d = np.zeros(2, dtype=np.float32), np.ones(2, dtype=np.int32)
    with pytest.raises(ValueError, match="mixture classes"):
        O.bayesr_sweep(X, xpx, y.copy(), a, d, 1.0, 0.2, [0.5, 0.5], 1, 1)
    with pytest.raises(ValueError, match="sum to 1"):
        O.bayesr_sweep(X, xpx, y.copy(), a, d, 1.0, 0.2, [0.5, 0.1, 0.1, 0.1], 1, 1)
    with pytest.raises(ValueError, match="one row per marker"):
        O.bayesr_sweep(X, xpx, y.copy(), a, d, 1.0, 0.2, np.full((3, 4), 0.25), 1, 1)
    with pytest.raises(ValueError, match="sigmaSq"):
        O.bayesr_sweep(X, xpx, y.copy(), a, d, 1.0, 0.0, [0.95, 0.03, 0.015, 0.005], 1, 1)


# ---- test/unit/test_annotated_bayesc.jl:473-490 ------------------------------------------------------
def test_bayesabc_rejects_mismatched_pi_vector():
    X = np.asfortranarray(np.eye(2, dtype=np.float32))
    z = np.zeros(2, dtype=np.float32)
    with pytest.raises(ValueError, match=r"pi vector length 1 must match the number of markers \(2\)"):
        O.bayesabc_sweep(X, np.ones(2, dtype=np.float32), np.array([0.2, -0.1], dtype=np.float32), z.copy(), z.copy(),
                         np.ones(2, dtype=np.float32), 1.0, np.ones(2), np.array([0.5]), 1, 1)


# ---- A.1 edge cases: pi = 0 includes everything, pi = 1 nothing -----------
def test_bayesc_pi_edge_cases(small_data):
    X, y = small_data["X"], small_data["y"]
    xpx = O.xpx(X)
    for pi, expect in ((0.0, X.shape[1]), (1.0, 0)):
        a = np.zeros(X.shape[1], dtype=np.float32)
        b, d = a.copy(), a.copy()
        O.bayesabc_sweep(X, xpx, (y - y.mean()).copy(), a, b, d, 0.5, 0.002, pi, 3, 1)
        assert int(d.sum()) == expect
        assert (a != 0).sum() == expect
        assert np.all(b != 0)          # a normal is drawn for beta even when excluded (BayesABC.jl:54)


# ---- the block form with one pass is the non-block chain (benchmarks/reports/2026-03-20-...:38-52) ----
@pytest.mark.parametrize("bs", [64, 256])
def test_block_one_pass_equals_dense_chain(small_data, bs):
    X, y = small_data["X"], small_data["y"]
    xpx = O.xpx(X)
    p = X.shape[1]
    bstarts = O.block_starts_for(p, bs)
    grams = O.grams_for(X, bstarts)
    out = []
    for blk in (False, True):
        r = (y - y.mean()).copy()
        a, b, d = (np.zeros(p, dtype=np.float32) for _ in range(3))
        kw = dict(block_starts=bstarts, grams=grams, nreps=1) if blk else {}
        for it in range(1, 31):
            O.bayesabc_sweep(X, xpx, r, a, b, d, 0.5, 0.004, 0.9, 2026, it, **kw)
        out.append((a, d, r))
    assert np.array_equal(out[0][1], out[1][1])
    assert np.abs(out[0][0] - out[1][0]).max() < 5e-6          # the reference saw 5e-8 mean |alpha| drift
    assert np.abs(out[0][2] - out[1][2]).max() < 5e-5


# ---- the one-block LOOKAHEAD schedule (what the HIP path runs, and what every GPU parity test compares with) is the
# ---- literal per-marker dot/axpy chain for the headline single-trait samplers (BayesABC.jl:60-80, BayesR.jl:45-97)
@pytest.mark.parametrize("method", ["BayesC", "BayesR"])
@pytest.mark.parametrize("bs", [64, 256])
def test_lookahead_equals_dense_chain(small_data, method, bs):
    X, y = small_data["X"], small_data["y"]
    xpx = O.xpx(X)
    p = X.shape[1]
    bstarts = O.block_starts_for(p, bs)
    grams = O.grams_for(X, bstarts)
    out = []
    for form in ("dense", "block", "lookahead"):
        r = (y - y.mean()).copy()
        a, b = (np.zeros(p, dtype=np.float32) for _ in range(2))
        d = np.ones(p, dtype=np.int32) if method == "BayesR" else np.zeros(p, dtype=np.float32)
        kw = {} if form == "dense" else dict(block_starts=bstarts, grams=grams, nreps=1, lookahead=(form == "lookahead"))
        for it in range(1, 31):
            if method == "BayesR":
                O.bayesr_sweep(X, xpx, r, a, d, 0.5, 0.05, np.array([0.9, 0.06, 0.03, 0.01]), 2026, it, **kw)
            else:
                O.bayesabc_sweep(X, xpx, r, a, b, d, 0.5, 0.004, 0.9, 2026, it, **kw)
        out.append((a, d, r))
        assert np.abs(r - ((y - y.mean()) - X.astype(np.float64) @ a.astype(np.float64))).max() < 2e-4    # residual identity
    assert (out[0][1] != (1 if method == "BayesR" else 0)).sum() > 5                                     # a non-trivial chain
    for o in out[1:]:
        assert np.array_equal(out[0][1], o[1])                  # indicator / class trajectories identical over 30 sweeps
        assert np.abs(out[0][0] - o[0]).max() < 5e-6            # the reference saw 5e-8 mean |alpha| drift block vs dense
        assert np.abs(out[0][2] - o[2]).max() < 5e-5


def test_f32_and_f64_accumulation_agree_within_fp32_noise(small_data):
    X, y = small_data["X"], small_data["y"]
    p = X.shape[1]
    res = []
    for acc in (O.ACC_F64, O.ACC_F32):
        xpx = O.xpx(X, acc)
        r = (y - y.mean()).copy()
        a, b, d = (np.zeros(p, dtype=np.float32) for _ in range(3))
        O.bayesabc_sweep(X, xpx, r, a, b, d, 0.5, 0.004, 0.9, 5, 1, acc=acc)
        res.append((a, d))
    assert (res[0][1] != res[1][1]).sum() <= 1
    m = res[0][1] == res[1][1]
    assert np.abs(res[0][0][m] - res[1][0][m]).max() < 1e-4


# ---- test/unit/test_multitrait_mcmc.jl:6-31,557-642: one-marker 2-trait closed-form posterior --------
def _exact_mt_state_probs(x, ys, vare, var_effect, prior):
    xp = float(x @ x)
    Rinv, Ginv = np.linalg.inv(vare), np.linalg.inv(var_effect)
    w = np.array([x @ y for y in ys])
    logd = np.zeros(4)
    for s in range(4):
        D = np.diag([(s >> 0) & 1, (s >> 1) & 1]).astype(float)
        lhs = D @ Rinv @ D * xp + Ginv
        rhs = (Rinv @ D).T @ w
        ghat = np.linalg.solve(lhs, rhs)
        logd[s] = -0.5 * (np.log(np.linalg.det(lhs)) - rhs @ ghat) + np.log(prior[s])
    pr = np.exp(logd - logd.max())
    return pr / pr.sum()


@pytest.mark.parametrize("block", [False, True])
def test_mt_sampler1_matches_closed_form_state_posterior(block):
    x = np.array([1.0, -0.5, 0.75])
    ys = [np.array([0.8, -0.1, 0.3]), np.array([0.2, 0.6, -0.4])]
    vare = np.array([[1.0, 0.25], [0.25, 0.9]])
    var_effect = np.array([[0.7, 0.15], [0.15, 0.8]])
    # states indexed delta_1 + 2*delta_2: 00, 10, 01, 11 (annotated_bayesc_mt_state_keys order)
    prior = np.array([0.35, 0.20, 0.15, 0.30])
    exact = _exact_mt_state_probs(x, ys, vare, var_effect, prior)
    X = np.asfortranarray(x.astype(np.float32)[:, None])
    xpx = O.xpx(X)
    r = np.ascontiguousarray(np.stack(ys).astype(np.float32))
    a, b, d = (np.zeros((2, 1), dtype=np.float32) for _ in range(3))
    kw = dict(block_starts=np.array([0]), grams=O.gram(X, 0, 1).ravel(), nreps=1) if block else {}
    counts = np.zeros(4)
    niter, burn = 20000, 3000
    for it in range(1, niter + 1):
        O.mtbayesc_I_sweep(X, xpx, r, a, b, d, vare, var_effect, np.log(prior), 20260411, it, **kw)
        if it > burn:
            counts[int(d[0, 0]) + 2 * int(d[1, 0])] += 1
    emp = counts / counts.sum()
    assert np.abs(emp - exact).max() < 0.02


@pytest.mark.parametrize("block", [False, True])
def test_mt_sampler2_matches_closed_form_state_posterior(block):
    """Sampler II draws the joint state from its marginal full conditional (MTBayesABC.jl:129-210):
    same one-marker closed form as above, plus E[beta | state 11] = lhs^-1 rhs."""
    x = np.array([1.0, -0.5, 0.75])
    ys = [np.array([0.8, -0.1, 0.3]), np.array([0.2, 0.6, -0.4])]
    vare = np.array([[1.0, 0.25], [0.25, 0.9]])
    var_effect = np.array([[0.7, 0.15], [0.15, 0.8]])
    prior = np.array([0.35, 0.20, 0.15, 0.30])
    exact = _exact_mt_state_probs(x, ys, vare, var_effect, prior)
    X = np.asfortranarray(x.astype(np.float32)[:, None])
    xpx = O.xpx(X)
    r = np.ascontiguousarray(np.stack(ys).astype(np.float32))
    a, b, d = (np.zeros((2, 1), dtype=np.float32) for _ in range(3))
    kw = dict(block_starts=np.array([0]), grams=O.gram(X, 0, 1).ravel(), nreps=1) if block else {}
    counts = np.zeros(4)
    bsum, nb = np.zeros(2), 0
    for it in range(1, 12001):
        O.mt_sweep(O.MT_SAMPLER_II, X, xpx, r, a, b, d, vare, var_effect, np.log(prior), 20260411, it, **kw)
        st = int(d[0, 0]) + 2 * int(d[1, 0])
        counts[st] += 1
        assert np.array_equal(a[:, 0], b[:, 0] * d[:, 0])              # alpha = D * beta (:201)
        if st == 3:
            bsum += b[:, 0]; nb += 1
    assert np.abs(counts / counts.sum() - exact).max() < 0.02
    xp = float(x @ x)
    Rinv, Ginv = np.linalg.inv(vare), np.linalg.inv(var_effect)
    ghat = np.linalg.solve(Rinv * xp + Ginv, Rinv.T @ np.array([x @ y for y in ys]))
    assert np.abs(bsum / nb - ghat).max() < 0.05
    # the residual identity holds: r = y - x*alpha
    assert np.abs(r - (np.stack(ys) - np.outer(a[:, 0], x))).max() < 1e-5


@pytest.mark.parametrize("kind", ["II", "mega"])
def test_mt_block_and_lookahead_forms_equal_dense_chain(small_data, kind):
    X = small_data["X"][:, :600]
    n, p = X.shape
    rng = np.random.default_rng(3)
    Y = np.stack([small_data["y"], small_data["y"][::-1] + rng.normal(size=n).astype(np.float32)]).astype(np.float32)
    Y -= Y.mean(axis=1, keepdims=True)
    xpx = O.xpx(X)
    bstarts = O.block_starts_for(p, 64)
    grams = O.grams_for(X, bstarts)
    vare = np.array([[0.6, 0.1], [0.1, 0.8]], dtype=np.float32)
    vg = np.array([[0.004, 0.001], [0.001, 0.005]], dtype=np.float32)
    if kind == "mega":
        k, prior = O.MT_MEGA, np.array([0.9, 0.8])
        vare, vg = np.diag(np.diag(vare)), np.diag(np.diag(vg))
    else:
        k, prior = O.MT_SAMPLER_II, np.log(np.array([0.8, 0.05, 0.05, 0.1]))
    out = []
    for form in ("dense", "block", "lookahead"):
        r = np.ascontiguousarray(Y.copy())
        a, b, d = (np.zeros((2, p), dtype=np.float32) for _ in range(3))
        kw = {} if form == "dense" else dict(block_starts=bstarts, grams=grams, nreps=1, lookahead=(form == "lookahead"))
        for it in range(1, 13):
            O.mt_sweep(k, X, xpx, r, a, b, d, vare, vg, prior, 77, it, **kw)
        out.append((a, d, r))
        assert np.abs(r - (Y - (X.astype(np.float64) @ a.T.astype(np.float64)).T)).max() < 2e-4
    for o in out[1:]:
        assert np.array_equal(out[0][1], o[1])
        assert np.abs(out[0][0] - o[0]).max() < 2e-5
        assert np.abs(out[0][2] - o[2]).max() < 2e-4
    assert 0 < out[0][1].sum() < 2 * p


def test_mega_trait0_is_the_single_trait_chain(small_data):
    """megaBayesABC! = t independent BayesABC! sweeps (BayesABC.jl:1-8): trait 0 (draw slot 0) reproduces the
    single-trait oracle bit for bit."""
    X = small_data["X"][:, :500]
    n, p = X.shape
    y = small_data["y"] - small_data["y"].mean()
    Y = np.ascontiguousarray(np.stack([y, 0.5 * y[::-1]]).astype(np.float32))
    xpx = O.xpx(X)
    vare = np.diag([0.5, 0.7]).astype(np.float32)
    vg = np.diag([0.004, 0.002]).astype(np.float32)
    r = Y.copy()
    a, b, d = (np.zeros((2, p), dtype=np.float32) for _ in range(3))
    r1 = Y[0].copy()
    a1, b1, d1 = (np.zeros(p, dtype=np.float32) for _ in range(3))
    for it in range(1, 9):
        O.mt_sweep(O.MT_MEGA, X, xpx, r, a, b, d, vare, vg, np.array([0.9, 0.7]), 11, it)
        O.bayesabc_sweep(X, xpx, r1, a1, b1, d1, 0.5, 0.004, 0.9, 11, it)
    assert np.array_equal(a[0], a1) and np.array_equal(b[0], b1) and np.array_equal(d[0], d1)
    assert np.array_equal(r[0], r1)
    assert d[1].sum() > 0


# ---- independent blocks (test_misc_coverage.jl:211-227; docs/src/manual/block_bayesc.md:95-134) ----------
def test_independent_blocks_exact_when_blocks_are_orthogonal():
    """X_b'X_c = 0 for b != c  =>  the independent-block sweep IS the sequential block sweep."""
    rng = np.random.default_rng(5)
    n, bsz, nblk = 192, 64, 3
    X = np.zeros((n, bsz * nblk), dtype=np.float32)
    for b in range(nblk):                                   # disjoint row supports -> orthogonal column blocks
        X[b * 64:(b + 1) * 64, b * bsz:(b + 1) * bsz] = rng.standard_normal((64, bsz)).astype(np.float32)
    X = np.asfortranarray(X)
    y = rng.standard_normal(n).astype(np.float32)
    xpx = O.xpx(X)
    bstarts = O.block_starts_for(X.shape[1], bsz)
    grams = O.grams_for(X, bstarts)
    res = []
    for indep in (False, True):
        r = y.copy()
        a, b, d = (np.zeros(X.shape[1], dtype=np.float32) for _ in range(3))
        for it in range(1, 6):
            O.bayesabc_sweep(X, xpx, r, a, b, d, 0.5, 0.05, 0.7, 9, it, block_starts=bstarts, grams=grams, nreps=2,
                             independent=indep)
        res.append((a, d, r))
    assert np.array_equal(res[0][1], res[1][1]) and np.array_equal(res[0][0], res[1][0])
    assert np.abs(res[0][2] - res[1][2]).max() < 1e-5


@pytest.mark.parametrize("method", ["BayesC", "BayesR", "MT"])
def test_independent_blocks_keep_the_residual_identity(small_data, method):
    """r = y - X alpha after the deferred reconcile (BayesABC.jl:251-253), and the chain differs from the
    sequential one when blocks are correlated (it is an approximation)."""
    X = small_data["X"][:, :640]
    n, p = X.shape
    y = (small_data["y"] - small_data["y"].mean()).astype(np.float32)
    xpx = O.xpx(X)
    bstarts = O.block_starts_for(p, 128)
    grams = O.grams_for(X, bstarts)
    kw = dict(block_starts=bstarts, grams=grams, nreps=1)
    out = []
    for indep in (False, True):
        if method == "MT":
            r = np.ascontiguousarray(np.stack([y, 0.5 * y[::-1]]).astype(np.float32))
            y0 = r.copy()
            a, b, d = (np.zeros((2, p), dtype=np.float32) for _ in range(3))
            for it in range(1, 7):
                O.mt_sweep(O.MT_SAMPLER_I, X, xpx, r, a, b, d, np.array([[0.5, 0.1], [0.1, 0.6]]),
                           np.array([[0.004, 0.001], [0.001, 0.004]]), np.log([0.7, 0.1, 0.1, 0.1]), 3, it,
                           independent=indep, **kw)
            assert np.abs(r - (y0 - (X.astype(np.float64) @ a.T.astype(np.float64)).T)).max() < 2e-4
        else:
            r = y.copy()
            a = np.zeros(p, dtype=np.float32)
            if method == "BayesC":
                b, d = np.zeros(p, dtype=np.float32), np.zeros(p, dtype=np.float32)
                for it in range(1, 7):
                    O.bayesabc_sweep(X, xpx, r, a, b, d, 0.5, 0.004, 0.9, 3, it, independent=indep, **kw)
            else:
                d = np.ones(p, dtype=np.int32)
                for it in range(1, 7):
                    O.bayesr_sweep(X, xpx, r, a, d, 0.5, 0.05, np.array([0.9, 0.05, 0.03, 0.02]), 3, it,
                                   independent=indep, **kw)
            assert np.abs(r - (y - X.astype(np.float64) @ a.astype(np.float64))).max() < 2e-4
        out.append(a.copy())
    assert not np.array_equal(out[0], out[1])


# ---- test/unit/test_streaming_codec.jl:21-51, test_streaming_prepare_lowmem.jl:22-49: 2-bit codec ------
def _pack_2bit(raw, missing=9):
    """Marker-major packing of streaming_genotypes.jl:364-367,622-627 (test-side restatement)."""
    n, p = raw.shape
    stride = (n + 3) // 4
    out = np.zeros(p * stride, dtype=np.uint8)
    for j in range(p):
        for i in range(n):
            code = 3 if raw[i, j] == missing else int(raw[i, j])
            out[j * stride + (i >> 2)] |= code << ((i & 3) << 1)
    return out


@pytest.mark.parametrize("with_missing", [True, False])
def test_2bit_decode_matches_dense_centered_imputed(with_missing):
    raw = np.array([[0, 1, 2, 0], [1, 0, 1, 2], [2, 9 if with_missing else 1, 0, 1], [0, 2, 1, 0], [1, 1, 2, 2], [2, 0, 0, 1]], dtype=np.float64)
    n, p = raw.shape
    payload = _pack_2bit(raw)
    assert payload.size == p * ((n + 3) // 4)
    for j in range(p):
        col = raw[:, j]
        nonmiss = col != 9
        mu = np.float32(col[nonmiss].sum() / nonmiss.sum())
        dense = np.where(nonmiss, col, mu).astype(np.float32) - mu            # readgenotypes.jl:372-384
        dec = O.decode_marker_2bit(payload, n, j, mu, centered=True)
        np.testing.assert_allclose(dec, dense, atol=1e-5)
        # xpRinvx sidecar formula (streaming_genotypes.jl:283-285): sum v^2 - mu * sum v over non-missing
        ss = float((col[nonmiss] ** 2).sum() - mu * col[nonmiss].sum())
        assert float(dec @ dec) == pytest.approx(ss, abs=1e-5)
        raw_dec = O.decode_marker_2bit(payload, n, j, mu, centered=False)
        np.testing.assert_allclose(raw_dec, np.where(nonmiss, col, mu), atol=1e-6)


# ---- output.jl:568-577 running means ---------------------------------------------------------------
def test_running_posterior_means_match_direct_moments():
    rng = np.random.default_rng(0)
    p, K = 50, 7
    ma, ma2, md = (np.zeros(p, dtype=np.float32) for _ in range(3))
    A = rng.standard_normal((K, p)).astype(np.float32)
    D = (rng.random((K, p)) < 0.4).astype(np.float32)
    for k in range(K):
        O.accumulate(A[k], D[k], k + 1, ma, ma2, md)
    np.testing.assert_allclose(ma, A.mean(axis=0), atol=1e-6)
    np.testing.assert_allclose(ma2, (A.astype(np.float64) ** 2).mean(axis=0), atol=1e-5)
    np.testing.assert_allclose(md, D.mean(axis=0), atol=1e-6)
    # BayesR: model frequency is the running mean of delta > 1 (output.jl:572-574)
    cls = rng.integers(1, 5, size=(K, p)).astype(np.int32)
    md[:] = 0; ma[:] = 0; ma2[:] = 0
    for k in range(K):
        O.accumulate(A[k], cls[k], k + 1, ma, ma2, md)
    np.testing.assert_allclose(md, (cls > 1).mean(axis=0), atol=1e-6)


# ---- test/unit/test_misc_coverage.jl:211-227: orthogonal blocks make the independent-block mode exact --
def test_orthogonal_shards_reconcile_exactly():
    d1 = make_dataset(n=120, p=64, ncausal=3, seed=1)
    d2 = make_dataset(n=130, p=64, ncausal=3, seed=2)
    X = np.zeros((250, 128), dtype=np.float32, order="F")
    X[:120, :64] = d1["X"]
    X[120:, 64:] = d2["X"]                      # X_1' X_2 = 0
    y = np.concatenate([d1["y"], d2["y"]]).astype(np.float32)
    xpx = O.xpx(X)
    full_r = (y - y.mean()).copy()
    a, b, d = (np.zeros(128, dtype=np.float32) for _ in range(3))
    O.bayesabc_sweep(X, xpx, full_r, a, b, d, 0.5, 0.01, 0.8, 4, 1)
    snap = (y - y.mean()).copy()
    deltas, a_sh = [], np.zeros(128, dtype=np.float32)
    for lo, hi in ((0, 64), (64, 128)):
        Xs = np.asfortranarray(X[:, lo:hi])
        r = snap.copy()
        aa, bb, dd = (np.zeros(64, dtype=np.float32) for _ in range(3))
        O.bayesabc_sweep(Xs, xpx[lo:hi].copy(), r, aa, bb, dd, 0.5, 0.01, 0.8, 4, 1, marker0=lo)
        deltas.append(r - snap)
        a_sh[lo:hi] = aa
    assert np.array_equal(a_sh, a)              # same draws (global marker index), same decisions
    np.testing.assert_allclose(snap + deltas[0] + deltas[1], full_r, atol=1e-6)


def test_per_marker_covariance_oracle_reduces_to_the_shared_one():
    """Multi-trait BayesA/B restatement (one G per marker, MTBayesABC.jl:66): with the SAME matrix for every marker the
    chain must be the multi-trait BayesC chain, bit for bit, in the dense, block and lookahead forms."""
    from oracle_engine import OracleEngine
    data = make_dataset(n=150, p=90, ncausal=5, seed=3)
    t = 3
    rng = np.random.default_rng(0)
    A = rng.standard_normal((t, t)); vare = (A @ A.T / t + np.eye(t)).astype(np.float32)
    B = rng.standard_normal((t, t)); G = ((B @ B.T / t + np.eye(t)) * 0.01).astype(np.float32)
    lp = np.log(rng.dirichlet(np.ones(1 << t)))
    for form in ("dense", "block", "lookahead"):
        out = []
        for method in ("MTBayesC", "MTBayesB"):
            e = OracleEngine(form)
            e.load_dense(data["X"])
            e.setup_blocks(32)
            e.init_state(method, t)
            for k in range(t):
                e.set_residual(((1 + 0.2 * k) * (data["y"] - data["y"].mean())).astype(np.float32), k)
                e.set_state(k, delta=np.ones(e.p, dtype=np.float32))
            kw = dict(vare=vare, var_effect=G, log_prior_states=lp)
            if method == "MTBayesB":
                kw["var_effect_matrix"] = np.tile(G, (e.p, 1, 1))
            for it in range(1, 6):
                e.sweep(iteration=it, seed=5, **kw)
            out.append([e.get_state(k) for k in range(t)] + [e.get_residual(k) for k in range(t)])
        for a, b in zip(out[0], out[1]):
            if isinstance(a, tuple):
                assert all(np.array_equal(x, y) for x, y in zip(a, b)), form
            else:
                assert np.array_equal(a, b), form


@pytest.mark.parametrize("method", ["BayesC", "BayesR", "MTBayesC"])
def test_float64_oracle_tracks_the_float32_oracle(method):
    """The Float64 restatement (oracle/jwas_oracle_f64.c: the reference's kernels with T = Float64, runMCMC(double_precision=
    true)) and the Float32 oracle use the same counter RNG: on a small well-conditioned problem their chains make the same
    inclusion / class decisions and the effects agree to Float32 rounding -- the Float64 oracle is pinned to the pinned one."""
    from oracle_engine import OracleEngine, OracleEngine64
    d = make_dataset(n=260, p=420, ncausal=6, seed=12)
    y = d["y"] - d["y"].mean()
    t = 2 if method == "MTBayesC" else 1
    rng = np.random.default_rng(1)
    if method == "BayesC":
        kw = dict(vare=0.5, var_effect=0.004, pi=0.9)
    elif method == "BayesR":
        kw = dict(vare=0.5, var_effect=0.05, pi_classes=np.array([0.9, 0.05, 0.03, 0.02]))
    else:
        kw = dict(vare=np.array([[0.6, 0.1], [0.1, 0.5]]), var_effect=np.array([[0.004, 0.001], [0.001, 0.003]]),
                  log_prior_states=np.log(np.array([0.7, 0.1, 0.1, 0.1])))
    res = {}
    for tag, e, dt in (("f32", OracleEngine("dense"), np.float32), ("f64", OracleEngine64(), np.float64)):
        e.load_dense(d["X"].astype(dt)); e.setup_blocks(64); e.init_state(method, t)
        for k in range(t):
            e.set_residual(((1 + 0.2 * k) * y).astype(dt), k)
            e.set_state(k, delta=np.ones(e.p, dtype=np.int32 if method == "BayesR" else dt))
        kws = {k_: (np.asarray(v, dtype=dt) if k_ in ("vare", "var_effect") else v) for k_, v in kw.items()}
        for it in range(1, 16):
            e.sweep(iteration=it, seed=5, **kws)
        res[tag] = [e.get_state(k) for k in range(t)]
    for k in range(t):
        assert np.array_equal(res["f32"][k][2], res["f64"][k][2])
        np.testing.assert_allclose(res["f32"][k][0], res["f64"][k][0], atol=5e-6)


def test_sampler1_linear_form_stays_within_a_few_ulp_of_the_literal_order():
    """Rule L (csrc/sweep.hpp, oracle mt1_update): a marker that is and stays in the model for every trait takes its new
    effects from the linear form beta = A w + c of its t conditionals instead of the conditional-by-conditional order --
    the same conditional means and draws, another association.  With the rule switched off the oracle is the literal
    restatement of _MTBayesABC_samplerI! (MTBayesABC.jl:85-120); both chains must make the same inclusion decisions and
    their effects may differ only at Float32 rounding level."""
    import oracle as O
    from oracle_engine import OracleEngine
    d = make_dataset(n=300, p=260, ncausal=6, seed=4)
    y = d["y"] - d["y"].mean()
    t = 3
    rng = np.random.default_rng(2)
    A = rng.standard_normal((t, t)); B = rng.standard_normal((t, t))
    prior = np.full(1 << t, 1e-3); prior[-1] = 1.0; prior /= prior.sum()             # (nearly) every marker in the model
    kw = dict(vare=((A @ A.T / t + np.eye(t)) * 0.5).astype(np.float32), var_effect=((B @ B.T / t + np.eye(t)) * 0.003).astype(np.float32),
              log_prior_states=np.log(prior))
    res = {}
    try:
        for on in (1, 0):
            O.lib().orc_set_mt_linear_form(on)
            e = OracleEngine("dense")
            e.load_dense(d["X"]); e.setup_blocks(64); e.init_state("MTBayesC", t)
            for k in range(t):
                e.set_residual(((1 + 0.2 * k) * y).astype(np.float32), k)
                e.set_state(k, delta=np.ones(e.p, dtype=np.float32))
            for it in range(1, 9):
                e.sweep(iteration=it, seed=3, **kw)
            res[on] = [e.get_state(k) for k in range(t)]
    finally:
        O.lib().orc_set_mt_linear_form(1)
    changed = 0
    for k in range(t):
        assert np.array_equal(res[1][k][2], res[0][k][2])
        scale = np.abs(res[0][k][0]).max()
        np.testing.assert_allclose(res[1][k][0], res[0][k][0], atol=2e-5 * scale)
        changed += int((res[1][k][0] != res[0][k][0]).sum())
    assert changed > 0                       # the rule did apply (the two orders round differently somewhere)


def test_rule_d_stays_within_float32_rounding_of_the_literal_order():
    """Rule D (csrc/kernels.hpp AbcMarker::rule_d, oracle abc_update): under a uniform prior pi = 0 every marker is included
    whatever its rhs and its new effect is one fused multiply-add of the block rhs, alpha = fmaf(c1, x, c0), instead of the
    chain rhs -> gHat -> alpha of bayesabc_update_marker! (BayesABC.jl:36,39,46).  With the rule switched off the oracle is
    the literal restatement; the two chains (RR-BLUP and BayesA settings, every marker in the model every sweep) agree to
    Float32 rounding -- the chain is a contraction there, rounding differences do not grow."""
    import oracle as O
    from oracle_engine import OracleEngine
    d = make_dataset(n=300, p=280, ncausal=6, seed=9)
    y = (d["y"] - d["y"].mean()).astype(np.float32)
    rng = np.random.default_rng(0)
    for method, kw in (("BayesC", dict(vare=np.float32(0.6), var_effect=np.float32(0.002), pi=0.0)),
                       ("BayesB", dict(vare=np.float32(0.6), var_effect=np.float32(0.002), pi=0.0,
                                       var_effect_vec=rng.uniform(0.001, 0.01, 280).astype(np.float32)))):
        res = {}
        try:
            for on in (True, False):
                O.RULE_D = on
                e = OracleEngine("dense")
                e.load_dense(d["X"]); e.setup_blocks(64); e.init_state(method, 1); e.set_residual(y)
                for it in range(1, 13):
                    st = e.sweep(iteration=it, seed=3, **kw)
                    assert st["sum_delta"][0] == e.p
                res[on] = (e.get_state(0)[0], e.get_residual(0))
        finally:
            O.RULE_D = True
        scale = np.abs(res[False][0]).max()
        np.testing.assert_allclose(res[True][0], res[False][0], atol=3e-6 * max(scale, 1.0))
        np.testing.assert_allclose(res[True][1], res[False][1], atol=2e-4)
        assert (res[True][0] != res[False][0]).any()             # the rule did apply
    # a per-marker pi vector of zeros does NOT trigger the rule (nor does it on the device: pi_vec sweeps run the general kernel)
    e1, e2 = OracleEngine("dense"), OracleEngine("dense")
    out = []
    for e, pi in ((e1, 0.0), (e2, np.zeros(280))):
        e.load_dense(d["X"]); e.setup_blocks(64); e.init_state("BayesC", 1); e.set_residual(y)
        e.sweep(iteration=1, seed=3, vare=np.float32(0.6), var_effect=np.float32(0.002), pi=pi)
        out.append(e.get_state(0)[0])
    assert (out[0] != out[1]).any() and np.abs(out[0] - out[1]).max() < 1e-5


def test_rule_t_stays_within_float32_rounding_of_the_literal_chain():
    """Rule T (csrc/sampler_mt.hpp dense_big_mt_solve, oracle mt1_section_solve): a 64-marker section of a full 256-marker block in
    which every marker is in the model for every trait is evaluated as D = T y with the section's inverse T = (I + L)^-1 instead
    of the 64-step chain.  Against the LITERAL restatement of the block form (_MTBayesABC_samplerI!, MTBayesABC.jl:243-333: no
    linear form, no solve): the same inclusion decisions in every sweep, effects within 2e-5 of their scale after eight dense
    sweeps (the chain is a contraction in this regime: rounding differences do not grow).  With a prior that lets markers
    leave the model the solve takes EXCEPTIONS (a marker outside the model at entry, or one the verification finds leaving:
    its literal evaluation replaces its row of the solution and the rows behind it take a rank-t correction) and the
    agreement stays; with many of them the section falls back to the sequential chain."""
    import oracle as O
    from oracle_engine import OracleEngine
    t = 3
    d = make_dataset(n=700, p=3 * 256 + 40, ncausal=12, seed=8)
    y = d["y"] - d["y"].mean()
    rng = np.random.default_rng(3)
    A = rng.standard_normal((t, t)); B = rng.standard_normal((t, t))
    vare = ((A @ A.T / t + np.eye(t)) * 0.5).astype(np.float32)
    varg = ((B @ B.T / t + np.eye(t)) * 0.003).astype(np.float32)
    for leak, expect_exceptions, expect_fallback in ((1e-9, False, False), (2e-3, True, False), (6e-2, True, True)):
        prior = np.full(1 << t, leak); prior[-1] = 1.0; prior /= prior.sum()
        kw = dict(vare=vare, var_effect=varg, log_prior_states=np.log(prior))
        res = {}
        try:
            for tag in ("literal", "solve"):
                O.lib().orc_set_mt_linear_form(0 if tag == "literal" else 1)
                e = OracleEngine("lookahead")
                e.load_dense(d["X"]); e.setup_blocks(256); e.init_state("MTBayesC", t)
                for k in range(t):
                    e.set_residual(((1 + 0.2 * k) * y).astype(np.float32), k)
                    e.set_state(k, delta=np.ones(e.p, dtype=np.float32))
                O.section_solve_counts(reset=True)
                traj = []
                for it in range(1, 9):
                    e.sweep(iteration=it, seed=5, section_solve=(tag == "solve"), **kw)
                    traj.append(np.stack([e.get_state(k)[2] for k in range(t)]).copy())
                res[tag] = ([e.get_state(k) for k in range(t)], traj, O.section_solve_counts(), O.section_solve_exceptions())
        finally:
            O.lib().orc_set_mt_linear_form(1)
        assert res["literal"][2] == (0, 0) and res["solve"][2][0] > 0
        assert (res["solve"][2][1] > 0) == expect_fallback and (res["solve"][3] > 0) == expect_exceptions
        if not expect_exceptions:
            for a, b in zip(res["literal"][1], res["solve"][1]):
                assert np.array_equal(a, b)
            for k in range(t):
                scale = np.abs(res["literal"][0][k][0]).max()
                np.testing.assert_allclose(res["solve"][0][k][0], res["literal"][0][k][0], atol=2e-5 * scale)
                assert (res["solve"][0][k][0] != res["literal"][0][k][0]).any()
        else:
            # a leaky prior: single decisions may flip at rounding level (an MCMC chain is chaotic), so the first sweep -- one
            # pass over identical inputs -- is compared exactly and the rest statistically
            assert np.array_equal(res["literal"][1][0], res["solve"][1][0])
            agree = np.mean(res["literal"][1][-1] == res["solve"][1][-1])
            assert agree > 0.97, agree


def test_packed_order_right_hand_side_equals_the_decoded_dot_product():
    """The 2-bit packed update role forms x'R^-1 r with the centring factored out of the sum (oracle dot_xr; decode_marker!,
    streaming_genotypes.jl:978-1002): sum_i c_i w_i r_i - mu (R - M).  Against the dot product of the decoded Float32 column --
    what the dense path and the reference's streaming path compute -- within 1e-6 relative (the decoded matrix carries the
    rounding of fl32(code - mu)); missing codes, residual weights, centred and uncentred storage; and a whole BayesC chain in
    that order stays within the reference's own stream-vs-dense tolerance (1e-4, test_streaming_codec.jl:100,104) of the chain
    on the decoded matrix."""
    import oracle as O
    from oracle_engine import OracleEngine
    rng = np.random.default_rng(11)
    n, p = 333, 150
    codes = rng.integers(0, 3, size=(n, p)).astype(np.uint8)
    codes[rng.integers(0, n, 60), rng.integers(0, p, 60)] = 3
    miss = codes == 3
    means = np.array([codes[~miss[:, j], j].mean(dtype=np.float32) for j in range(p)], dtype=np.float32)
    y = rng.standard_normal(n).astype(np.float32)
    for centered in (True, False):
        v = np.where(miss, means[None, :], codes.astype(np.float32)).astype(np.float32)
        X = np.asfortranarray(v - means[None, :] if centered else v)
        for weights in (None, rng.uniform(0.5, 2.0, n).astype(np.float32)):
            res = {}
            for tag in ("decoded", "packed"):
                e = OracleEngine("lookahead")
                e.load_dense(X); e.set_weights(weights); e.setup_blocks(64); e.init_state("BayesC", 1)
                e.set_residual(y - y.mean())
                try:
                    if tag == "packed":
                        e.set_packed_source(codes, means, centered=centered)
                    for it in range(1, 7):
                        e.sweep(iteration=it, seed=2, vare=np.float32(0.8), var_effect=np.float32(0.01), pi=0.7)
                finally:
                    e.set_packed_source(None, None)
                res[tag] = (e.get_state(0), e.get_residual(0))
            scale = max(float(np.abs(res["decoded"][0][0]).max()), 1e-3)
            np.testing.assert_allclose(res["packed"][0][0], res["decoded"][0][0], rtol=0, atol=1e-4 * scale)
            assert (res["packed"][0][2] == res["decoded"][0][2]).mean() > 0.98
