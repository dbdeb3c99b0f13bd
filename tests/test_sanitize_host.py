"""CPU: the product's host code (host_calls.cpp, bam_host.cpp, hostmath.h, the generators) and the oracle built with
AddressSanitizer + UndefinedBehaviorSanitizer and driven by tests/sanitize/host_harness.cpp (sanitizers
on the CPU build only, never on the GPU).  The harness also compares the candidate stages' host path and the histogram
quantiles with the oracle, and walks truncated / bit-flipped copies of a BAM."""
import os
import shutil
import subprocess

import pytest

import bam_util as bu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.timeout(600)
def test_host_code_under_asan_ubsan(tmp_path):
    if shutil.which("g++") is None:
        pytest.skip("no g++")
    probe = subprocess.run(["g++", "-fsanitize=address,undefined", "-x", "c++", "-", "-o", str(tmp_path / "probe")], input=b"int main(){return 0;}",
                           capture_output=True)
    if probe.returncode != 0:
        pytest.skip("g++ cannot link the sanitizer runtimes here")
    r = subprocess.run(["make", "-f", "tests/sanitize/Makefile"], cwd=ROOT, capture_output=True, text=True)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    bam, _, _ = bu.build_golden_bam(str(tmp_path))
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=0:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")
    r = subprocess.run([os.path.join(ROOT, "tests", "sanitize", "host_harness"), bam], capture_output=True, text=True, env=env, timeout=500)
    assert r.returncode == 0 and "host harness ok" in r.stdout, (r.stdout[-1500:] + r.stderr[-3000:])


@pytest.mark.timeout(300)
def test_pool_gate_under_tsan(tmp_path):
    """The pool's turnstile for per-base phases (rsicnv_amd/csrc/gate.h) from twelve threads under ThreadSanitizer, both
    schedules, one to three turns at a time: no data race, never more turns at once than allowed, no shared section inside an
    exclusive turn."""
    if shutil.which("g++") is None:
        pytest.skip("no g++")
    probe = subprocess.run(["g++", "-fsanitize=thread", "-x", "c++", "-", "-o", str(tmp_path / "probe")], input=b"int main(){return 0;}",
                           capture_output=True)
    if probe.returncode != 0:
        pytest.skip("g++ cannot link the ThreadSanitizer runtime here")
    r = subprocess.run(["make", "-f", "tests/sanitize/Makefile", "tests/sanitize/gate_tsan"], cwd=ROOT, capture_output=True, text=True)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    r = subprocess.run([os.path.join(ROOT, "tests", "sanitize", "gate_tsan")], capture_output=True, text=True, timeout=250)
    assert r.returncode == 0 and "gate harness ok" in r.stdout and "ThreadSanitizer" not in r.stderr, (r.stdout[-1500:] + r.stderr[-3000:])


@pytest.mark.timeout(300)
def test_pool_run_queue_under_tsan(tmp_path):
    """The pool's queue of runs (rsicnv_amd/csrc/run_queue.h: rsi_pool_submit / rsi_pool_wait) under ThreadSanitizer: eleven
    workers, three clients submitting and waiting out of order and helping while they wait -- every item once, finish once per
    run and before its waiter returns, no helper on a younger run, no data race."""
    if shutil.which("g++") is None:
        pytest.skip("no g++")
    probe = subprocess.run(["g++", "-fsanitize=thread", "-x", "c++", "-", "-o", str(tmp_path / "probe")], input=b"int main(){return 0;}",
                           capture_output=True)
    if probe.returncode != 0:
        pytest.skip("g++ cannot link the ThreadSanitizer runtime here")
    r = subprocess.run(["make", "-f", "tests/sanitize/Makefile", "tests/sanitize/queue_tsan"], cwd=ROOT, capture_output=True, text=True)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    r = subprocess.run([os.path.join(ROOT, "tests", "sanitize", "queue_tsan")], capture_output=True, text=True, timeout=250)
    assert r.returncode == 0 and "queue harness ok" in r.stdout and "ThreadSanitizer" not in r.stderr, (r.stdout[-1500:] + r.stderr[-3000:])


@pytest.mark.timeout(300)
def test_pool_retirement_policy_under_tsan(tmp_path):
    """Who stops taking chromosomes when contexts are poisoned (rsicnv_amd/csrc/retire.h): with all but one threaded
    worker poisoned and runs submitted but not waited for the queue drains; with every worker poisoned the last claiming thread
    goes on; a pool of one keeps its only seat; nobody retires while nothing is poisoned."""
    if shutil.which("g++") is None:
        pytest.skip("no g++")
    probe = subprocess.run(["g++", "-fsanitize=thread", "-x", "c++", "-", "-o", str(tmp_path / "probe")], input=b"int main(){return 0;}",
                           capture_output=True)
    if probe.returncode != 0:
        pytest.skip("g++ cannot link the ThreadSanitizer runtime here")
    r = subprocess.run(["make", "-f", "tests/sanitize/Makefile", "tests/sanitize/retire_tsan"], cwd=ROOT, capture_output=True, text=True)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    r = subprocess.run([os.path.join(ROOT, "tests", "sanitize", "retire_tsan")], capture_output=True, text=True, timeout=250)
    assert r.returncode == 0 and "retire harness ok" in r.stdout and "ThreadSanitizer" not in r.stderr, (r.stdout[-1500:] + r.stderr[-3000:])
