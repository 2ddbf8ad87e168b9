"""`python bench.py --gpus N` started WITHOUT a launcher starts its own ranks (torch.distributed.run as a child
process), rank 0's single JSON line comes through, a failing rank makes the exit code non-zero.  CAPHN_BENCH_DRYRUN=1
replaces the GPU work with a sleep, so the control flow (spawn, gloo rendezvous on 127.0.0.1, max over ranks, one
line) runs here on CPU; tests/test_gpu_bench.py runs the real thing on the GPU box."""
import json
import os
import subprocess
import sys

import pytest

from conftest import REPO


def _run(extra_env, *args, timeout=240):
    env = dict(os.environ)
    env.pop("WORLD_SIZE", None); env.pop("RANK", None); env.pop("LOCAL_RANK", None)
    env.update(extra_env)
    return subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), *args], env=env, capture_output=True, text=True,
                          timeout=timeout)


@pytest.mark.timeout(300)
def test_bare_gpus_2_launches_two_ranks_and_prints_one_line():
    r = _run({"CAPHN_BENCH_DRYRUN": "1"}, "--gpus", "2", "--steps", "3", "--warmup", "1")
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 3 and out["warmup"] == 1 and out["dryrun"] is True
    assert out["max_over_ranks_s"] >= 0.02          # rank 1 sleeps 20 ms: the reported time is the slowest rank's


@pytest.mark.timeout(300)
def test_a_failing_rank_fails_the_launcher():
    r = _run({"CAPHN_BENCH_DRYRUN": "1", "CAPHN_BENCH_DRYRUN_FAIL_RANK": "1"}, "--gpus", "2", "--steps", "1", "--warmup", "0")
    assert r.returncode != 0
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith('{"metric"')]


def test_more_ranks_than_devices_is_refused_with_a_message():
    import torch
    if torch.cuda.device_count() >= 2:
        pytest.skip("needs a box with fewer than 2 GPUs")
    r = _run({}, "--gpus", "2", "--steps", "1", "--warmup", "0", timeout=120)
    assert r.returncode == 2 and "needs 2 GPUs" in r.stderr
