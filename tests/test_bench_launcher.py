"""CPU: ``bench.py --gpus N`` starts N rank processes itself (no WORLD_SIZE in the environment), joins them over
gloo in --dry-run mode (stand-in forward: there is no GPU here) and reports what actually ran."""
import json
import os
import subprocess
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(*argv, env_extra=None):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT",
                                                            "TORCHELASTIC_RUN_ID")}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), *argv], env=env, capture_output=True, text=True, timeout=600)


def _json_line(stdout):
    lines = [ln for ln in stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, stdout            # rank 0 prints ONE line, the other ranks none
    return json.loads(lines[0])


def _both_ranks_profiled(r, line):
    """The profiled pass re-runs a step that holds collectives: both ranks must go through it (rank 0 alone would wait for
    partners that never come) and rank 0 reports the records."""
    for rk in (0, 1):
        assert f"rank {rk}/2: profiled pass ran (step holds collectives" in r.stderr, r.stderr[-2000:]
    assert "profiled_ms_per_step" in line and line["roofline"]["kernel"] == "dry_run_step"


def test_gpus_2_launches_two_ranks_batch_sharded():
    r = _run("--gpus", "2", "--dry-run", "--steps", "2", "--warmup", "1")
    assert r.returncode == 0, r.stderr[-2000:]
    line = _json_line(r.stdout)
    assert line["n_gpus"] == 2 and line["dry_run"] is True and line["scaling"] == "weak"
    assert line["config"]["parallelism"] == "batch-sharded x2"
    assert "rank 1/2" in r.stderr and "rank 0/2" in r.stderr        # both ranks really ran
    # whole-job value: both ranks' frames over the MAX time
    assert abs(line["value"] - 2 * 8 * 128 * 128 / 1e6 / (line["ms_per_step"] * 1e-3)) / line["value"] < 1e-3
    # collective-free step: rank 0 profiles alone
    assert "rank 0/2: profiled pass ran" in r.stderr and "rank 1/2: profiled pass ran" not in r.stderr


def test_cfg4_two_ranks_tile_sharded_with_the_all_gather_in_the_step():
    r = _run("--gpus", "2", "--dry-run", "--workload", "cfg4t", "--steps", "2", "--warmup", "1")
    assert r.returncode == 0, r.stderr[-2000:]
    line = _json_line(r.stdout)
    assert line["n_gpus"] == 2 and line["scaling"] == "strong"
    assert "tile-sharded x2" in line["config"]["parallelism"] and "all-gather" in line["config"]["parallelism"]
    assert ", 2 tiles" in r.stderr
    _both_ranks_profiled(r, line)


@pytest.mark.parametrize("name", ["cfg4", "cfg4x"])          # BASELINE configs[3] runs in the exact mode by default
def test_cfg4_two_ranks_exact_row_shards_with_their_collectives_in_the_step(name):
    r = _run("--gpus", "2", "--dry-run", "--workload", name, "--steps", "2", "--warmup", "1")
    assert r.returncode == 0, r.stderr[-2000:]
    line = _json_line(r.stdout)
    assert line["n_gpus"] == 2 and line["scaling"] == "strong"
    assert "exact row shards x2" in line["config"]["parallelism"] and "all-reduced" in line["config"]["parallelism"]
    _both_ranks_profiled(r, line)


def test_cfg5_two_ranks_rehearse_the_bucketed_gradient_all_reduce():
    r = _run("--gpus", "2", "--dry-run", "--workload", "cfg5", "--steps", "2", "--warmup", "1")
    assert r.returncode == 0, r.stderr[-2000:]
    line = _json_line(r.stdout)
    assert line["n_gpus"] == 2 and line["scaling"] == "weak" and "training step" in line["metric"]
    assert "all-reduce" in line["config"]["parallelism"]
    _both_ranks_profiled(r, line)


def test_world_size_mismatch_is_refused():
    r = _run("--gpus", "2", "--dry-run", env_extra={"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0 and "refusing" in (r.stderr + r.stdout)


def test_single_rank_without_a_gpu_fails_loudly():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    r = _run("--steps", "1", "--warmup", "0")
    assert r.returncode != 0 and "no CPU fallback" in (r.stderr + r.stdout)
