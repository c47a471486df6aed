"""Committed golden vectors (tests/golden/sweep_golden.npz, made by tests/golden/make_golden.py):
the oracle must reproduce them on CPU (regression pin); the HIP path must match them on the GPU."""
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
import make_golden as MG  # noqa: E402
from oracle_engine import OracleEngine  # noqa: E402


def _check(engine, name):
    g = np.load(os.path.join(HERE, "golden", "sweep_golden.npz"))
    X = MG.centered(g["raw"])
    case = MG.CASES[name]
    res = MG.run_case(engine, X, [g["y1"], g["y2"], MG.third_trait(g["y1"], g["y2"])], case)
    for k, v in res.items():
        ref = g[f"{name}/{k}"]
        if k.startswith("delta") or k == "n_events_last":
            assert np.array_equal(v, ref), (name, k)
        else:
            np.testing.assert_allclose(v, ref, rtol=0, atol=2e-6 if k.startswith(("alpha", "beta")) else 2e-5, err_msg=f"{name}/{k}")


@pytest.mark.parametrize("name", list(MG.CASES))
def test_oracle_reproduces_golden(name):
    _check(OracleEngine("lookahead"), name)


@pytest.mark.gpu
@pytest.mark.parametrize("name", list(MG.CASES))
def test_hip_matches_golden(name):
    import jwas_jl_amd as J
    e = J.HipEngine(0)
    try:
        _check(e, name)
    finally:
        e.close()
