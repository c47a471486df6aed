"""Randomised small shapes: ragged n (not a multiple of 256 or 4), p smaller than / not a multiple of the block size,
single-block and single-row-group cases, every sampler -- HIP path vs the oracle's lookahead restatement."""
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


CASES = [
    # (n, p, block, method)
    (1, 70, 64, "BayesC"), (3, 5, 64, "BayesC"), (5, 64, 64, "BayesC"), (255, 65, 64, "BayesR"), (257, 129, 128, "BayesC"),
    (1023, 300, 256, "BayesR"), (2049, 1025, 1024, "BayesC"), (2304, 513, 512, "MTBayesC"), (77, 200, 128, "MTBayesC_II"),
    (301, 1000, 512, "MegaBayesC"), (4097, 90, 64, "BayesB"), (600, 2047, 1024, "BayesR"),
]


@pytest.mark.parametrize("n,p,bs,method", CASES)
def test_ragged_shapes(hip, n, p, bs, method):
    rng = np.random.default_rng(n * 1000 + p)
    d = make_dataset(n=max(n, 2), p=p, ncausal=min(5, p), seed=n + p)
    X = np.asfortranarray(d["X"][:n])
    y = d["y"][:n].astype(np.float32)
    y = y - y.mean() if n > 1 else y
    t = 2 if method in ("MTBayesC", "MTBayesC_II", "MegaBayesC") else 1
    orc = OracleEngine("lookahead")
    for e in (orc, hip):
        e.load_dense(X)
        e.setup_blocks(bs, "f64")
        e.init_state(method, t)
        for k in range(t):
            e.set_residual((1 + 0.3 * k) * y, k)
        if method == "BayesR":
            e.set_state(0, delta=np.ones(p, dtype=np.int32))
    v = np.float32(max(float(np.var(y)), 0.1))
    g = np.float32(0.02)
    if method == "BayesC":
        kw = dict(vare=v, var_effect=g, pi=0.8)
    elif method == "BayesB":
        kw = dict(vare=v, var_effect=g, var_effect_vec=rng.uniform(0.01, 0.03, p).astype(np.float32), pi=0.7)
    elif method == "BayesR":
        kw = dict(vare=v, var_effect=np.float32(0.1), pi_classes=np.array([0.85, 0.08, 0.04, 0.03]))
    elif method == "MegaBayesC":
        kw = dict(vare=np.diag([v, 2 * v]).astype(np.float32), var_effect=np.diag([g, g]).astype(np.float32), pi=np.array([0.8, 0.7]))
    else:
        kw = dict(vare=np.array([[v, 0.1 * v], [0.1 * v, 1.5 * v]], dtype=np.float32),
                  var_effect=np.array([[g, 0.3 * g], [0.3 * g, g]], dtype=np.float32),
                  log_prior_states=np.log(np.array([0.7, 0.1, 0.1, 0.1])))
    for it in range(1, 7):
        so = orc.sweep(iteration=it, seed=99, **kw)
        sh = hip.sweep(iteration=it, seed=99, **kw)
        assert so["n_events"] == sh["n_events"], f"iteration {it}"
    for k in range(t):
        ao, bo, do = orc.get_state(k)
        ah, bh, dh = hip.get_state(k)
        assert np.array_equal(do, dh)
        np.testing.assert_allclose(ah, ao, rtol=0, atol=1e-5)
        np.testing.assert_allclose(hip.get_residual(k), orc.get_residual(k), rtol=0, atol=5e-5)
    np.testing.assert_allclose(sh["resid_ss"], so["resid_ss"], rtol=1e-5, atol=1e-6)


def _random_case(seed):
    """One random configuration of the sweep: shape, block partition (uniform size or explicit ragged starts), sampler,
    prior sparsity (dense ... very sparse), repetitions (1, a few, 0 = block size), traits."""
    rng = np.random.default_rng(seed)
    method = rng.choice(["BayesC", "BayesC", "BayesR", "BayesB", "MTBayesC", "MTBayesC_II", "MegaBayesC", "MTBayesB"])
    t = 1 if method in ("BayesC", "BayesR", "BayesB") else int(rng.integers(2, 4))
    pervar = method == "MTBayesB"                                       # multi-trait BayesA/B: a covariance per marker ...
    if pervar:                                                          # ... under either Gibbs sampler, or constrained
        method = ("MTBayesB", "MTBayesB_II", "MegaBayesB")[int(seed) % 3]
    n = int(rng.integers(40, 700)) if rng.random() < 0.85 else int(rng.integers(1500, 4200))      # (the tall ones: several row groups)
    p = int(rng.integers(30, 900))
    explicit = rng.random() < 0.4
    if explicit:
        cuts = np.unique(rng.integers(1, p, size=int(rng.integers(1, 12))))
        starts = np.concatenate([[0], cuts]).astype(np.int64)
        while np.diff(np.append(starts, p)).max() > 300:                # keep blocks small enough for the oracle's patience
            big = int(np.argmax(np.diff(np.append(starts, p))))
            starts = np.sort(np.append(starts, starts[big] + np.diff(np.append(starts, p))[big] // 2))
        part = ("explicit", starts)
    else:
        part = ("uniform", int(rng.choice([64, 128, 256, 512])))
    nreps = int(rng.choice([1, 1, 1, 2, 5, 0]))
    if nreps == 0:
        # block size = repetition count: keep it <= 128 (the oracle's patience; the 512-repetition case is pinned in
        # test_every_bit_equal_when_the_oracle_sums_in_the_device_order)
        if part[0] == "uniform":
            part = ("uniform", min(part[1], 128))
        else:
            starts = part[1]
            while np.diff(np.append(starts, p)).max() > 128:
                big = int(np.argmax(np.diff(np.append(starts, p))))
                starts = np.sort(np.append(starts, starts[big] + np.diff(np.append(starts, p))[big] // 2))
            part = ("explicit", starts)
    if pervar and not explicit and part[1] * t > 2048:
        part = ("uniform", 256)
    sparsity = float(rng.choice([0.0, 0.3, 0.9, 0.99]))
    weights = rng.random() < 0.25                                       # heterogeneous residuals (x'R^-1 x, X_b'R^-1 r)
    # independent_blocks=true (on explicit partitions: every fifth seed -- no extra draw, so the other cases keep their configuration)
    indep = (not pervar) and ((int(seed) % 5 == 0) if explicit else rng.random() < 0.2)
    marker_prior = method in ("BayesC", "BayesR") and rng.random() < 0.25       # per-marker pi (annotation priors)
    coop = rng.random() < 0.3                                           # cooperative dense apply forced on
    return dict(method=method, t=t, n=n, p=p, part=part, nreps=nreps, sparsity=sparsity, seed=int(seed),
                weights=weights, indep=indep, marker_prior=marker_prior, coop=coop)


@pytest.mark.parametrize("seed", list(range(int(__import__("os").environ.get("JWAS_FUZZ_CASES", "400")))))     # JWAS_FUZZ_CASES=6000 for a long run
def test_random_configurations_against_the_oracle(hip, seed, monkeypatch):
    """Differential fuzzing of the whole configuration space (a seeded, reproducible sample of it): whatever the shape,
    partition, sampler, prior, repetition count, residual weights, independent blocks, per-marker priors and update-role
    variant, the device chain equals the oracle's BIT FOR BIT (effects, indicators, residuals)."""
    c = _random_case(1000 + seed)
    monkeypatch.setenv("JWAS_HIP_COOP_APPLY", "1" if c["coop"] else "0")
    rng = np.random.default_rng(c["seed"])
    method, t, n, p = c["method"], c["t"], c["n"], c["p"]
    d = make_dataset(n=n, p=p, ncausal=min(6, p), seed=c["seed"] % 1000)
    y = (d["y"] - d["y"].mean()).astype(np.float32)
    import oracle as O
    w = rng.uniform(0.3, 3.0, n).astype(np.float32) if c["weights"] else None
    hip.load_dense(d["X"]); hip.set_weights(w)
    # the oracle sums a block's x'r in the device's association order and both sides use the oracle's precomputed inner
    # products: the comparison below is bit for bit (see test_every_bit_equal_when_the_oracle_sums_in_the_device_order)
    O.set_device_order(hip.update_geometry()[0])
    orc = OracleEngine("lookahead", acc=O.ACC_DEVICE)
    orc.load_dense(d["X"]); orc.set_weights(w)
    for e in (orc, hip):
        if c["part"][0] == "explicit":
            e.setup_blocks_explicit(c["part"][1], "f64")
        else:
            e.setup_blocks(c["part"][1], "f64")
    hip.set_xpx(orc._xpx); hip.set_grams_packed(orc._grams)
    orc._w()
    st_ = list(orc._bs) + [p]
    for kb in range(1, len(st_) - 1):
        hip.set_cross_gram(kb, O.cross_gram(d["X"], st_[kb - 1], st_[kb] - st_[kb - 1], st_[kb], st_[kb + 1] - st_[kb], O.ACC_DEVICE))
    O.set_weights(None)
    for e in (orc, hip):
        e.init_state(method, t)
        for k in range(t):
            e.set_residual(((1 + 0.3 * k) * y).astype(np.float32), k)
        if method == "BayesR":
            e.set_state(0, delta=np.ones(p, dtype=np.int32))
        elif t > 1:
            for k in range(t):
                e.set_state(k, delta=np.ones(p, dtype=np.float32))
    v = np.float32(max(float(np.var(y)), 0.1))
    g = np.float32(0.02)
    sp = c["sparsity"]
    A = rng.standard_normal((t, t)); Rm = ((A @ A.T / t + np.eye(t)) * v).astype(np.float32)
    Bm = rng.standard_normal((t, t)); Gm = ((Bm @ Bm.T / t + np.eye(t)) * g).astype(np.float32)
    if method == "BayesC":
        kw = dict(vare=v, var_effect=g, pi=sp)
        if c["marker_prior"]:
            kw = dict(vare=v, var_effect=g, pi_vec=np.clip(sp + rng.uniform(-0.2, 0.2, p), 0.0, 0.999))
    elif method == "BayesB":
        kw = dict(vare=v, var_effect=g, var_effect_vec=rng.uniform(0.005, 0.04, p).astype(np.float32), pi=sp)
    elif method == "BayesR":
        rest = np.array([0.5, 0.3, 0.2]) * (1 - sp)
        kw = dict(vare=v, var_effect=np.float32(0.1), pi_classes=np.concatenate([[sp], rest]))
        if c["marker_prior"]:
            pm = rng.dirichlet(np.ones(4), size=p) * 0.5 + 0.5 * kw["pi_classes"]
            kw["pi_matrix"] = pm / pm.sum(axis=1, keepdims=True)
    elif method == "MegaBayesC":
        kw = dict(vare=np.diag(np.diag(Rm)), var_effect=np.diag(np.diag(Gm)), pi=np.full(t, sp))
    elif method == "MegaBayesB":                                        # (the marker's own diagonal variances)
        Vm = np.zeros((p, t, t), dtype=np.float32)
        for k in range(t):
            Vm[:, k, k] = g * np.exp(rng.uniform(-1, 1, p))
        kw = dict(vare=np.diag(np.diag(Rm)), var_effect=np.eye(t, dtype=np.float32), pi=np.full(t, sp), var_effect_matrix=Vm)
    else:
        prior = np.full(1 << t, (1 - sp) / ((1 << t) - 1)); prior[0] = sp
        if sp == 0.0:
            prior = np.full(1 << t, 1e-3); prior[-1] = 1.0
        prior /= prior.sum()
        kw = dict(vare=Rm, var_effect=Gm, log_prior_states=np.log(prior))
        if method in ("MTBayesB", "MTBayesB_II"):
            Wm = rng.standard_normal((p, t, t))
            kw["var_effect_matrix"] = ((Wm @ Wm.transpose(0, 2, 1) / t + np.eye(t)) * (g * np.exp(rng.uniform(-1, 1, p)))[:, None, None]).astype(np.float32)
    nreps = c["nreps"] if method not in ("MTBayesC_II", "MTBayesB_II") or c["nreps"] in (1, 2) else 1
    if c["indep"]:
        kw["independent_blocks"] = True
    for it in range(1, 5):
        so = orc.sweep(iteration=it, seed=c["seed"], nreps=nreps, **kw)
        sh = hip.sweep(iteration=it, seed=c["seed"], nreps=nreps, **kw)
        assert so["n_events"] == sh["n_events"], f"{c} iteration {it}"
    O.set_device_order(8)
    for k in range(t):
        ao, bo, do = orc.get_state(k)
        ah, bh, dh = hip.get_state(k)
        assert np.array_equal(do, dh), f"{c}"
        assert np.array_equal(ah, ao) and np.array_equal(bh, bo), f"{c}: max |d alpha| {np.abs(ah - ao).max()}"
        assert np.array_equal(hip.get_residual(k), orc.get_residual(k)), f"{c}"


@pytest.mark.parametrize("seed", list(range(max(40, int(__import__("os").environ.get("JWAS_FUZZ_CASES", "400")) // 8))))
def test_random_rule_t_configurations_against_the_oracle(hip, seed):
    """Differential fuzzing of RULE T (jwas_sweep_params.section_solve: dense 64-marker sections of multi-trait sampler I as
    triangular solves with per-sweep section inverses, csrc/sampler_mt.hpp): random shapes (one to four full 256-marker blocks
    and a ragged tail), two or three traits, a shared or a per-marker effect covariance, priors from "nothing ever leaves the
    model" to "most sections collect more exceptions than the rule takes", a few or many markers outside the model at the start
    (exceptions of a solved section / sections that are walked), residual weights -- with the oracle's sums in the device's
    order and the oracle's Grams on the device, the chains are equal BIT FOR BIT, and the same sections are solved / fall back,
    with the same number of exceptions, on both sides."""
    import oracle as O
    rng = np.random.default_rng(77_000 + seed)
    t = int(rng.integers(2, 4))
    method = "MTBayesB" if rng.random() < 0.35 else "MTBayesC"
    n = int(rng.integers(300, 900)) if rng.random() < 0.7 else int(rng.integers(1500, 3300))
    p = 256 * int(rng.integers(1, 5)) + int(rng.choice([0, 0, 17, 130, 255]))
    leak = float(rng.choice([0.0, 1e-9, 1e-4, 3e-3, 2e-2, 6e-2]))
    d = make_dataset(n=n, p=p, ncausal=min(10, p), seed=int(rng.integers(0, 1000)))
    y = (d["y"] - d["y"].mean()).astype(np.float32)
    w = rng.uniform(0.3, 3.0, n).astype(np.float32) if rng.random() < 0.25 else None
    hip.load_dense(d["X"]); hip.set_weights(w)
    O.set_device_order(hip.update_geometry()[0])
    orc = OracleEngine("lookahead", acc=O.ACC_DEVICE)
    orc.load_dense(d["X"]); orc.set_weights(w)
    for e in (orc, hip):
        e.setup_blocks(256, "f64")
    hip.set_xpx(orc._xpx); hip.set_grams_packed(orc._grams)
    orc._w()
    st_ = list(orc._bs) + [p]
    for kb in range(1, len(st_) - 1):
        hip.set_cross_gram(kb, O.cross_gram(d["X"], st_[kb - 1], st_[kb] - st_[kb - 1], st_[kb], st_[kb + 1] - st_[kb], O.ACC_DEVICE))
    O.set_weights(None)
    d0 = np.ones((t, p), dtype=np.float32)
    if rng.random() < 0.4:                                              # a few markers start outside the model for a trait
        k_out = int(rng.integers(1, 12)) if rng.random() < 0.7 else int(rng.integers(12, p // 8))
        d0[rng.integers(0, t, k_out), rng.choice(p, k_out, replace=False)] = 0.0
    for e in (orc, hip):
        e.init_state(method, t)
        for k in range(t):
            e.set_residual(((1 + 0.3 * k) * y).astype(np.float32), k)
            e.set_state(k, delta=d0[k])
    v = np.float32(max(float(np.var(y)), 0.1))
    A = rng.standard_normal((t, t)); Rm = ((A @ A.T / t + np.eye(t)) * v).astype(np.float32)
    Bm = rng.standard_normal((t, t)); Gm = ((Bm @ Bm.T / t + np.eye(t)) * 0.02).astype(np.float32)
    prior = np.full(1 << t, leak); prior[-1] = 1.0; prior /= prior.sum()
    with np.errstate(divide="ignore"):
        kw = dict(vare=Rm, var_effect=Gm, log_prior_states=np.log(prior), section_solve=True)
    if method == "MTBayesB":
        Wm = rng.standard_normal((p, t, t))
        kw["var_effect_matrix"] = ((Wm @ Wm.transpose(0, 2, 1) / t + np.eye(t)) * (0.02 * np.exp(rng.uniform(-1, 1, p)))[:, None, None]).astype(np.float32)
    O.section_solve_counts(reset=True)
    solved = fallen = exceptions = 0
    try:
        for it in range(1, 5):
            so = orc.sweep(iteration=it, seed=1000 + seed, **kw)
            sh = hip.sweep(iteration=it, seed=1000 + seed, **kw)
            cnt = hip.last_sweep_counters()
            solved += cnt[16]; fallen += cnt[17]; exceptions += cnt[23]
            assert so["n_events"] == sh["n_events"], f"seed {seed} iteration {it}"
    finally:
        O.set_device_order(8)
    assert (solved, fallen, exceptions) == O.section_solve_counts() + (O.section_solve_exceptions(),), f"seed {seed}"
    for k in range(t):
        ao, bo, do = orc.get_state(k)
        ah, bh, dh = hip.get_state(k)
        assert np.array_equal(do, dh), f"seed {seed}"
        assert np.array_equal(ah, ao) and np.array_equal(bh, bo), f"seed {seed}: max |d alpha| {np.abs(ah - ao).max()}"
        assert np.array_equal(hip.get_residual(k), orc.get_residual(k)), f"seed {seed}"


@pytest.mark.parametrize("seed", list(range(max(24, int(__import__("os").environ.get("JWAS_FUZZ_CASES", "400")) // 8))))
def test_random_multitrait_skip_and_verify_configurations_against_the_oracle(hip, seed):
    """Differential fuzzing of the multi-trait SKIP AND VERIFY pass (csrc/sampler_mt.hpp: helper waves evaluate the 64-marker
    sub-blocks without a candidate, the serial wave takes the chain over again when one of their markers moves): random shapes (one to
    three blocks of 256 / 512 / 1024 markers and a ragged tail), two to four traits, samplers I / II / constraint = true / a covariance per
    marker, sparse priors of different strength, copies of the causal columns a few sub-blocks behind the originals (they make skipped
    markers move), residual weights -- bit for bit against the oracle summing in the device's order."""
    import oracle as O
    rng = np.random.default_rng(91_000 + seed)
    method = str(rng.choice(["MTBayesC", "MTBayesC", "MTBayesC_II", "MegaBayesC", "MTBayesB"]))
    bs = int(rng.choice([256, 512, 512, 1024]))
    t = int(rng.integers(2, 5)) if method != "MTBayesB" else int(rng.integers(2, 4))
    while method == "MTBayesB" and bs * t > 2048:
        bs //= 2
    n = int(rng.integers(300, 900))
    p = bs * int(rng.integers(1, 4)) + int(rng.choice([0, 17, 130, bs // 2 + 3]))
    d = make_dataset(n=n, p=p, ncausal=min(24, p), h2=0.7, seed=int(rng.integers(0, 1000)))
    X = d["X"].copy()
    if rng.random() < 0.6:
        for j in d["causal"]:
            k = int(j) + int(rng.choice([64, 130, 200, 330]))
            if k < p and k // bs == int(j) // bs:
                X[:, k] = X[:, j]
    X = np.asfortranarray(X)
    y = (d["y"] - d["y"].mean()).astype(np.float32)
    w = rng.uniform(0.3, 3.0, n).astype(np.float32) if rng.random() < 0.25 else None
    hip.load_dense(X); hip.set_weights(w)
    O.set_device_order(hip.update_geometry()[0])
    orc = OracleEngine("lookahead", acc=O.ACC_DEVICE)
    orc.load_dense(X); orc.set_weights(w)
    for e in (orc, hip):
        e.setup_blocks(bs, "f64")
    hip.set_xpx(orc._xpx); hip.set_grams_packed(orc._grams)
    orc._w()
    st_ = list(orc._bs) + [p]
    for kb in range(1, len(st_) - 1):
        hip.set_cross_gram(kb, O.cross_gram(X, st_[kb - 1], st_[kb] - st_[kb - 1], st_[kb], st_[kb + 1] - st_[kb], O.ACC_DEVICE))
    O.set_weights(None)
    for e in (orc, hip):
        e.init_state(method, t)
        for k in range(t):
            e.set_residual(((1 + 0.3 * k) * y + 0.3 * np.random.default_rng(seed * 7 + k).standard_normal(n)).astype(np.float32), k)
    v = np.float32(max(float(np.var(y)), 0.1))
    A = rng.standard_normal((t, t)); Rm = ((A @ A.T / t + np.eye(t)) * v * 0.5).astype(np.float32)
    Bm = rng.standard_normal((t, t)); Gm = ((Bm @ Bm.T / t + np.eye(t)) * 0.01).astype(np.float32)
    sp = float(rng.choice([0.9, 0.97, 0.995]))
    if method == "MegaBayesC":
        kw = dict(vare=np.diag(np.diag(Rm)).astype(np.float32), var_effect=np.diag(np.diag(Gm)).astype(np.float32), pi=np.full(t, 1 - (1 - sp) / t))
    else:
        prior = np.full(1 << t, (1 - sp) / ((1 << t) - 1)); prior[0] = sp
        kw = dict(vare=Rm, var_effect=Gm, log_prior_states=np.log(prior))
        if method == "MTBayesB":
            Wm = rng.standard_normal((p, t, t))
            kw["var_effect_matrix"] = ((Wm @ Wm.transpose(0, 2, 1) / t + np.eye(t)) * 0.01).astype(np.float32)
    try:
        for it in range(1, 9):
            so = orc.sweep(iteration=it, seed=2000 + seed, **kw)
            sh = hip.sweep(iteration=it, seed=2000 + seed, **kw)
            assert so["n_events"] == sh["n_events"], f"seed {seed} iteration {it} ({method}, t = {t}, block {bs}, p = {p})"
    finally:
        O.set_device_order(8)
    for k in range(t):
        ao, bo, do = orc.get_state(k)
        ah, bh, dh = hip.get_state(k)
        assert np.array_equal(do, dh), f"seed {seed}"
        assert np.array_equal(ah, ao) and np.array_equal(bh, bo), f"seed {seed}: max |d alpha| {np.abs(ah - ao).max()}"
        assert np.array_equal(hip.get_residual(k), orc.get_residual(k)), f"seed {seed}"
