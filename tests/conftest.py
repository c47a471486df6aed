import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.dirname(os.path.abspath(__file__))):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def make_dataset(n=500, p=2000, ncausal=20, h2=0.5, seed=2026, center=True, dtype=np.float32):
    """SURVEY.md section 8d config-1 generator (the shape of benchmarks/bayesr_parity_common.jl:34-41):
    f_j ~ U(0.1,0.4), x_ij = Bernoulli(f_j)+Bernoulli(f_j), centred; ncausal N(0,1) effects scaled to
    heritability h2; y = 1 + X beta + e."""
    rng = np.random.default_rng(seed)
    f = rng.uniform(0.1, 0.4, size=p)
    X = (rng.random((n, p)) < f).astype(dtype) + (rng.random((n, p)) < f).astype(dtype)
    means = X.mean(axis=0, dtype=np.float64).astype(dtype)
    raw = X.copy()
    if center:
        X = X - means
    X = np.asfortranarray(X.astype(np.float32))
    causal = rng.choice(p, size=ncausal, replace=False)
    beta = np.zeros(p)
    beta[causal] = rng.standard_normal(ncausal)
    g = X.astype(np.float64) @ beta
    g *= np.sqrt(h2 / max(g.var(), 1e-30))
    e = rng.standard_normal(n) * np.sqrt(1.0 - h2)
    y = (1.0 + g + e).astype(np.float32)
    return {"X": X, "raw": raw, "means": means, "y": y, "causal": causal, "freq": means / 2.0}


@pytest.fixture(scope="session")
def small_data():
    return make_dataset(n=300, p=640, ncausal=10, seed=11)


@pytest.fixture(scope="session")
def config1_data():
    return make_dataset(n=500, p=2000, ncausal=20, seed=2026)
