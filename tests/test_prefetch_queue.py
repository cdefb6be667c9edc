"""CPU tests of the bookkeeping of ma_amd/host/ma_engine.h's PrefetchQueue (the reader-side funnel of the drop-in graph,
export.cpp:99-126) with a stand-in engine: tests/emul/prefetch_queue_test.cpp."""
import json
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "tests", "emul", "prefetch_queue_test")


@pytest.fixture(scope="module")
def exe():
    src = EXE + ".cpp"
    deps = [src, os.path.join(ROOT, "ma_amd", "host", "ma_engine.h"), os.path.join(ROOT, "include", "ma_amd.h")]
    if not os.path.exists(EXE) or any(os.path.getmtime(d) > os.path.getmtime(EXE) for d in deps):
        subprocess.check_call(["g++", "-std=c++17", "-O2", "-Wall", "-I" + os.path.join(ROOT, "include"), src, "-o", EXE, "-lpthread"])
    return EXE


@pytest.mark.parametrize("mode", ["two", "eof", "replicas", "abort", "throw"])
def test_prefetch_queue_bookkeeping(exe, mode):
    """two: the same threads serve two queues alternately -- every read exactly once, tickets lead to the read's own record
    (one thread-local slice per thread dropped reads); eof: a source is never asked again after its end marker;
    replicas: device batches rotate over three index replicas; abort: a destroyed queue's slices are never handed out and its
    results are released; throw: a non-std exception of the source fails every caller instead of hanging them."""
    out = subprocess.run([exe, mode], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr
    rec = json.loads(out.stdout.strip().splitlines()[-1])
    if mode in ("two", "eof", "replicas"):
        assert rec["seen_once"] == rec["reads"] and rec["bad_tickets"] == 0
    if mode == "replicas":
        assert min(rec["runs_per_replica"]) > 0


def test_prefetch_queue_is_race_free_under_thread_sanitizer(tmp_path):
    """The same five scenarios with the queue compiled under -fsanitize=thread: several graph threads feed one PrefetchQueue, seal batches, wait for tickets, a queue is destroyed under
    them, a source throws -- no data race reported, the same bookkeeping results."""
    exe = str(tmp_path / "prefetch_queue_tsan")
    p = subprocess.run(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=thread", "-Wall", "-I" + os.path.join(ROOT, "include"), EXE + ".cpp", "-o", exe, "-lpthread"],
                       capture_output=True, text=True)
    if p.returncode != 0:
        pytest.skip("no ThreadSanitizer runtime: " + p.stderr[-300:])
    for mode in ("two", "eof", "replicas", "abort", "throw"):
        out = subprocess.run([exe, mode], capture_output=True, text=True, timeout=300)
        assert out.returncode == 0, out.stdout + out.stderr
        assert "ThreadSanitizer" not in out.stderr, out.stderr[-3000:]
        rec = json.loads(out.stdout.strip().splitlines()[-1])
        if mode in ("two", "eof", "replicas"):
            assert rec["seen_once"] == rec["reads"] and rec["bad_tickets"] == 0
