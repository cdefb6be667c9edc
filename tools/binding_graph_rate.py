#!/usr/bin/env python3
"""Throughput of the per-read graph on the reference's REAL types (oracle/_ref/ref_graph_test = the reference's own
promiseMe / Pledge / simultaneousGet / FileWriter, the five ma_amd:: modules of ma_ref_binding.h): reader node wrapped into
ma_amd::PrefetchReader against the per-read funnel and against the reference's own CPU modules.  Needs the compiled
reference (oracle/_ref) and a GPU.   usage: python tools/binding_graph_rate.py [reads=200000] > gpurun_out/binding_graph_rate.txt"""
import json
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from ma_testlib import rand_genome, sample_reads, write_case  # noqa: E402

EXE = os.path.join(ROOT, "oracle", "_ref", "ref_graph_test")


def run(case, out, stages, threads, options, env=None):
    e = dict(os.environ, **(env or {}))
    o = subprocess.check_output([EXE, "sam", case, "default", "1", out, stages, str(threads), str(options)], env=e).decode()
    return json.loads(o.strip().splitlines()[-1])


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 200000
    g = rand_genome(1, [4641652])
    reads = sample_reads(g, n, 150, 11, sub=0.005)
    with tempfile.TemporaryDirectory() as td:
        case = os.path.join(td, "c.case")
        write_case(case, g, reads)
        out = os.path.join(td, "o.sam")
        print("%d x 150 bp reads vs a 4.6 Mnt genome, the reference's graph runtime and FileWriter; reads/s of simultaneousGet" % n)
        want = None
        for threads in (4, 8, 16, 32):
            r = run(case, out, "all", threads, 8, {"MA_PREFETCH_BATCH": "65536"})
            lines = sorted(open(out).read().splitlines())
            want = want or lines
            assert lines == want
            print("ma_amd:: modules + PrefetchReader, %2d graph threads: %10.0f reads/s (%d device batches, %d reads through the funnel)" % (
                threads, r["reads_per_s"], r["prefetched_batches"], r["reads_in_batches"]))
        for threads in (8, 16, 32):  # the same graph with ma_amd::BufferedFileWriter (the reference's formatter, its lock once per 64 KB)
            r = run(case, out, "all", threads, 8 | 32, {"MA_PREFETCH_BATCH": "65536"})
            assert sorted(open(out).read().splitlines()) == want
            print("ma_amd:: modules + PrefetchReader + BufferedFileWriter, %2d graph threads: %10.0f reads/s" % (threads, r["reads_per_s"]))
        for threads in (256, 1024):
            r = run(case, out, "all", threads, 0)
            assert sorted(open(out).read().splitlines()) == want
            print("ma_amd:: modules, per-read funnel, %4d graph threads: %10.0f reads/s (%d device batches)" % (threads, r["reads_per_s"], r["device_batches"]))
        m = min(n, 20000)
        write_case(case, g, reads[:m])
        r = run(case, out, "none", 1, 0)
        print("the reference's own CPU modules, 1 graph thread (its Harmonization draws from the process-wide rand()), %d reads: %10.0f reads/s" % (
            m, r["reads_per_s"]))
        assert sorted(open(out).read().splitlines()) == sorted(l for l in want if l.split("\t")[0] in set("r%d" % i for i in range(m)) or l.startswith("@"))


if __name__ == "__main__":
    main()
