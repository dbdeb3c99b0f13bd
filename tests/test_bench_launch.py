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


@pytest.mark.gpu
@pytest.mark.timeout(1500)
def test_sharded_genome_at_full_size_gives_the_reference_rows():
    """the sharded run at FULL size against the reference's rows (tests/golden/genome_rows.json), with 2 and
    with 4 ranks over gloo on the one GPU of the box (more ranks than that would exceed the box's limit of processes on its
    card), and once more with the host cut to two cores per rank -- eight ranks on a sixteen-core node -- where the pool must
    size itself to the cores (rsi.cpp:2189-2217 is the loop being sharded; its writer, rsi.cpp:1592-1616, orders the rows)."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["RSI_BENCH_BACKEND"] = "gloo"
    common = ["--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--no-single"]
    record = {}
    for n, cpus in ((2, 0), (4, 0), (4, 2)):
        e = dict(env)
        if cpus:
            e["RSI_BENCH_CPUS_PER_RANK"] = str(cpus)
        d = _line(["--gpus", str(n)] + common, e)
        cfg = d["config"]
        assert d["n_gpus"] == n and cfg["world_size"] == n and cfg["backend"] == "gloo" and d["scaling"] == "strong"
        assert d["rows_match_reference"] is True and d["steps_identical"], (n, cpus)
        bases = [r["bases"] for r in cfg["ranks"]]
        assert sum(bases) == cfg["genome_bases"]
        assert sum(r["chromosomes"] for r in cfg["ranks"]) == 24
        assert max(bases) <= 1.08 * (sum(bases) / n), bases          # longest-first to the least loaded rank
        if cpus:
            assert cfg["cores_per_rank"] == cpus and cfg["workers"] == max(4, 2 * cpus)
        else:
            assert cfg["workers"] == min(16, max(4, 2 * cfg["cores_per_rank"]))      # (20 on one GPU, 16 per rank of several)
        record[f"{n} ranks" + (f", {cpus} cores per rank" if cpus else "")] = {
            "ms_per_step": d["ms_per_step"], "workers": cfg["workers"], "cores_per_rank": cfg["cores_per_rank"], "steps_in_flight": cfg["steps_in_flight"],
            "rows_match_reference": d["rows_match_reference"], "bases_per_rank": bases}
    out = os.path.join(ROOT, "gpurun_out")
    if os.path.isdir(out):
        with open(os.path.join(out, "r5_sharded_rehearsal.json"), "w") as f:
            json.dump({"note": "bench.py --gpus N over gloo, all ranks on ONE MI355X (a rehearsal of the sharded path, not a scaling curve)", "runs": record}, f, indent=1)
