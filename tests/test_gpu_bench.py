"""bench.py on the GPU box: the bare `--gpus 2` launch (rehearsal: both ranks on the one GPU, collectives over gloo),
and the one-rank RCCL run with forced collectives.  Short runs: the control flow and the JSON contract, not the numbers."""
import json
import os
import subprocess
import sys

import pytest

from conftest import REPO

pytestmark = pytest.mark.gpu


def _bench(env_extra, *args, timeout=420):
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    env.update(env_extra)
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), *args], env=env, capture_output=True, text=True,
                       timeout=timeout)
    assert r.returncode == 0, (r.stdout[-1000:], r.stderr[-3000:])
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout
    return json.loads(lines[0])


@pytest.mark.timeout(600)
def test_bare_gpus_2_rehearsal_prints_one_line_for_two_ranks():
    out = _bench({"CAPHN_BENCH_REHEARSAL": "1"}, "--gpus", "2", "--steps", "3", "--warmup", "1", "--no-spinup", "--batch", "16")
    assert out["n_gpus"] == 2 and out["config"]["global_batch"] == 32 and out["scaling"] == "weak"
    assert out["value"] > 0 and "REHEARSAL" in out["config"]["parallelism"]
    assert "cpu_baseline" not in out            # rank 0 at N = 1 only


@pytest.mark.timeout(600)
def test_single_rank_rccl_group_with_forced_collectives():
    out = _bench({"CAPHN_FORCE_COLLECTIVES": "1"}, "--gpus", "1", "--steps", "5", "--warmup", "2", "--no-spinup",
                 "--no-cpu-baseline")
    assert out["n_gpus"] == 1 and "collectives forced" in out["config"]["parallelism"]
    assert out["roofline"]["frac"] > 0.3 and out["config"]["final_loss"] < 9.5
