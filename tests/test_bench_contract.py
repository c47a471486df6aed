"""bench.py's launch contract (no GPU needed): `python bench.py --gpus N` run plainly must start N ranks itself or fail --
it may never print a line for fewer GPUs than it was asked for; under a launcher whose WORLD_SIZE differs from --gpus it
refuses as well."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, env_extra=None, timeout=300):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, capture_output=True, text=True, timeout=timeout)


def test_bench_refuses_more_gpus_than_the_box_has():
    import torch
    ndev = torch.cuda.device_count()
    want = max(ndev, 1) + 1
    r = _run(["--gpus", str(want), "--steps", "1", "--warmup", "1", "--no-cpu-baseline"])
    assert r.returncode == 2, r.stderr[-2000:]
    assert f"needs {want} GPUs" in r.stderr
    assert "{" not in r.stdout                       # no JSON line


def test_bench_refuses_a_world_size_that_is_not_gpus():
    r = _run(["--gpus", "4", "--steps", "1", "--warmup", "1"], {"RANK": "0", "WORLD_SIZE": "2", "LOCAL_RANK": "0"})
    assert r.returncode == 2, r.stderr[-2000:]
    assert "refusing to print a line" in r.stderr
    assert "{" not in r.stdout


@pytest.mark.gpu
def test_bench_self_spawn_two_ranks_on_one_device_debug_path():
    """The self-spawn path end to end on a one-GPU box: `python bench.py --gpus 2` with the debug switches that let two
    ranks share device 0 over gloo (the exchange goes through the host).  The JSON line must say n_gpus = 2 and carry
    one device sweep time per rank."""
    import json
    r = _run(["--gpus", "2", "--steps", "2", "--warmup", "2", "--burnin", "0", "--n", "3000", "--p", "8192", "--no-cpu-baseline"],
             {"JWAS_BENCH_ONE_DEVICE": "1", "JWAS_BENCH_BACKEND": "gloo"}, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1]
    out = json.loads(line)
    assert out["n_gpus"] == 2 and len(out["config"]["per_rank_device_sweep_ms"]) == 2
    assert out["config"]["parallelism"].startswith("marker-shard x2")


@pytest.mark.gpu
def test_bench_row_shards_single_rank_equals_the_plain_chain():
    """--shard rows with one rank (a real one-rank RCCL communicator): the exact chain, same statistics as the plain run."""
    import json
    outs = []
    for extra in ([], ["--shard", "rows"]):
        r = _run(["--steps", "3", "--warmup", "3", "--burnin", "0", "--n", "3000", "--p", "8192", "--no-cpu-baseline", "--via-api", "0"] + extra, timeout=900)
        assert r.returncode == 0, r.stderr[-3000:]
        outs.append(json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1]))
    assert outs[0]["config"]["markers_in_model"] == outs[1]["config"]["markers_in_model"]
    assert outs[0]["config"]["events_per_sweep"] == outs[1]["config"]["events_per_sweep"]


@pytest.mark.gpu
def test_bench_one_rank_communicator_runs_the_library_sharded_path():
    """--one-rank-comm (round 4: what one rank of an N-GPU job executes per iteration, profiles/r04_rank_share.json): the sweep goes
    through jwas_hip_sweep_sharded on a ONE-rank RCCL communicator -- snapshot, pack kernel, ncclAllReduce, apply kernel -- and
    yields the plain run's chain (a one-term sum changes nothing)."""
    import json
    outs = []
    for extra in ([], ["--one-rank-comm"]):
        r = _run(["--steps", "3", "--warmup", "3", "--burnin", "0", "--n", "3000", "--p", "8192", "--no-cpu-baseline", "--via-api", "0"] + extra, timeout=900)
        assert r.returncode == 0, r.stderr[-3000:]
        outs.append(json.loads([ln for ln in r.stdout.splitlines() if ln.startswith('{"metric"')][-1]))
    assert outs[1]["config"]["sharded_path"] is True and outs[0]["config"]["sharded_path"] is False
    assert outs[0]["config"]["markers_in_model"] == outs[1]["config"]["markers_in_model"]
    assert outs[0]["config"]["events_per_sweep"] == outs[1]["config"]["events_per_sweep"]
    assert "host_ms_per_step" in outs[1]["config"]
