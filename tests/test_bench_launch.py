"""bench.py's multi-rank line cannot come out wrong: `--gpus N` without a launcher starts the N ranks
itself, a WORLD_SIZE that disagrees with --gpus is an error, and a two-rank run (gloo, both ranks on the one GPU of the test
box) reports n_gpus = 2 and the very rows of the one-rank run."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def test_world_size_that_disagrees_with_gpus_is_refused():
    """No GPU needed: the check sits in front of the torch import."""
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "1", "--warmup", "0"], capture_output=True, text=True, env=env, timeout=120)
    assert r.returncode != 0 and "WORLD_SIZE=1 but --gpus 2" in (r.stderr + r.stdout)
    assert not r.stdout.strip().startswith("{")


def _line(args, env):
    r = subprocess.run([sys.executable, BENCH] + args, capture_output=True, text=True, env=env, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    return json.loads(lines[0])


@pytest.mark.gpu
@pytest.mark.timeout(1200)
def test_two_ranks_without_a_launcher_report_two_gpus_and_the_same_rows():
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["RSI_BENCH_BACKEND"] = "gloo"
    common = ["--scale", "0.01", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-single", "--workers", "6"]
    two = _line(["--gpus", "2"] + common, env)
    one = _line(["--gpus", "1"] + common, env)
    assert two["n_gpus"] == 2 and two["config"]["world_size"] == 2 and two["config"]["backend"] == "gloo"
    assert [r["rank"] for r in two["config"]["ranks"]] == [0, 1]
    assert sum(r["chromosomes"] for r in two["config"]["ranks"]) == 24 and sum(r["bases"] for r in two["config"]["ranks"]) == two["config"]["genome_bases"]
    assert one["n_gpus"] == 1 and one["config"]["world_size"] == 1
    assert two["steps_identical"] and one["steps_identical"]
    assert two["rows_sha256"] == one["rows_sha256"] and two["config"]["calls_per_genome"] == one["config"]["calls_per_genome"] > 0
    assert two["rows_match_reference"] is None      # a scaled genome has no reference rows: the check must not claim any
