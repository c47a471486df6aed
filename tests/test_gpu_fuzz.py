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
