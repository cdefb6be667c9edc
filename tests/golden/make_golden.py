#!/usr/bin/env python3
"""Generates the golden vectors in this directory from the REAL reference.

Needs oracle/_ref/ref_dump (built by `make -C oracle ref` from /root/reference; only possible in the
build container).  The outputs are data only: synthetic inputs (case files) and the reference's
results on them (text dumps / index files).  Run:  python tests/golden/make_golden.py
"""
import gzip
import os
import shutil
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from ma_testlib import (sample_inversion_reads, sample_pairs, rand_genome, sample_reads, write_case, write_ksw_cases, rand_ksw_cases, run_ref, have_ref)  # noqa


def gz(path):
    with open(path, "rb") as f, gzip.open(path + ".gz", "wb", compresslevel=9) as g:
        shutil.copyfileobj(f, g)
    os.remove(path)


SAM_GOLDENS = [("default", 0), ("default", 1), ("default", 2), ("default", 3), ("illumina", 0), ("default", 4), ("default", 5),
               ("illumina", 4)]


def sam_goldens():
    """SAM text of the reference's FileWriter (fileWriter.cpp:11-158) for small.case; options: bit 0 = soft clip,
    bit 1 = '='/'X' cigars instead of 'M', bit 2 = "Emulate NGMLR's tag output" (MD SV AS NM XI XE XR CV SA QS QE)."""
    for preset, opt in SAM_GOLDENS:
        name = "small_ref.%s.opt%d.sam" % (preset, opt)
        run_ref("sam", "small.case", preset, 1, name, opt)
        gz(name)


READER_INPUTS = {
    "reader_multi.fa": ">r1 some description here\nACGTACGTNNacgt\nGGGTTT\n\n>r2\nAC\r\nGT\r\n>r3_iupac extra\nRYKMSWACGT*\n ACGT\n>r4\nTTTT",
    "reader_multi.fq": "@q1 desc\nACGTN\nACG\n+\nIIIII\nIII\n@q2\nGGGG\n+\n@@@@\n@q3\nACGTACGT\n+\nIIIIIIII",
    "reader_empty.fa": ">e1\n\n>e2\nACGT\n",
    "reader_notfasta.txt": "hello\n>x\nACGT\n",
    "reader_plusname.fq": "@a\nACGT\n+a\nIIII\n",
}


def reader_goldens():
    """What the reference's FileReader (fileReader.cpp:37-196) returns for small FASTA/FASTQ inputs, and the SAM text of
    its FileWriter when the reads (names, qualities) come from a FASTQ file."""
    os.makedirs("reader", exist_ok=True)
    for name, text in READER_INPUTS.items():
        with open(os.path.join("reader", name), "w", newline="") as f:
            f.write(text)
        run_ref("read", os.path.join("reader", name), os.path.join("reader", name + ".ref"))
    # FASTQ with qualities for the first 24 reads of small.case
    from ma_testlib import read_case
    _, reads, _ = read_case("small.case")
    with open(os.path.join("reader", "small24.fq"), "w") as f:
        for i, r in enumerate(reads[:24]):
            seq = "".join("ACGTN"[int(c)] for c in r)
            qual = "".join(chr(33 + (7 * i + 3 * k) % 40) for k in range(len(seq)))
            f.write("@fq%d sample=%d\n%s\n+\n%s\n" % (i, i, seq, qual))
    run_ref("sam", "small.case", "default", 1, os.path.join("reader", "small24.fq.sam"), 0, os.path.join("reader", "small24.fq"))
    gz(os.path.join("reader", "small24.fq.sam"))
    # gzip-compressed inputs (GzFileStream of the reference's WITH_ZLIB build, oracle/Makefile.ref)
    for name in ("reader_multi.fa", "reader_multi.fq", "reader_plusname.fq"):
        with open(os.path.join("reader", name), "rb") as f, gzip.open(os.path.join("reader", name + ".gz"), "wb", compresslevel=9) as g:
            shutil.copyfileobj(f, g)
        run_ref("read", os.path.join("reader", name + ".gz"), os.path.join("reader", name + ".gz.ref"))
    # mate files for the PairedFileReader: second file one record shorter (EOF of either ends the pairs), Ns in a mate
    with open(os.path.join("reader", "mates_1.fq"), "w") as f:
        f.write("@p0/1\nACGTTGCA\n+\nIIIIHHHH\n@p1/1\nGGGTTTAACC\n+\n0123456789\n@p2/1\nAAAA\n+\n!!!!\n")
    with open(os.path.join("reader", "mates_2.fq"), "w") as f:
        f.write("@p0/2 x\nTTGACCNA\n+\nABCDEFGH\n@p1/2\nCATG\n+\n9876\n")
    for rc in (0, 1):
        run_ref("readpair", os.path.join("reader", "mates_1.fq"), os.path.join("reader", "mates_2.fq"),
                os.path.join("reader", "mates.rc%d.ref" % rc), rc)


KSW_SCORINGS = [(3, 5, 6, 3, 30, 2), (1, 3, 5, 2, 24, 1), (2, 4, 24, 1, 4, 2), (5, 4, 2, 1, 40, 1)]


def ksw_scoring_goldens():
    """G7b: the kswcpp cases of ksw.case under other (match, mismatch, gap, extend, gap2, extend2) than the presets'."""
    made = not os.path.exists("ksw.case")
    if made:
        with gzip.open("ksw.case.gz", "rb") as g, open("ksw.case", "wb") as f:
            shutil.copyfileobj(g, f)
    for k, sc in enumerate(KSW_SCORINGS):
        run_ref("ksw", "ksw.case", "ksw_ref.sc%d.out" % k, "clean", *sc)
        gz("ksw_ref.sc%d.out" % k)
    if made:
        os.remove("ksw.case")


# (preset, search inversions, paired, Z Drop Inversions, SAM options) of the f4 goldens
F4_CONFIGS = [("default", 1, 0, 100, 0), ("default", 1, 1, 100, 0), ("illumina", 0, 1, 100, 3), ("default", 1, 1, 40, 1)]


def f4_name(preset, inv, paired, zd, opt):
    return "f4.%s.inv%d.pair%d.zd%d.opt%d" % (preset, inv, paired, zd, opt)


def f4_goldens():
    """f4: SmallInversions + PairedReads + PairedFileWriter of the reference on mate pairs and reads with small inversions."""
    g = rand_genome(101, [30000, 22000, 9000], repeat_unit=200, repeat_copies=25, repeat_div=0.06)  # = small.case
    reads = (sample_pairs(g, 120, 150, 401) + sample_pairs(g, 10, 250, 402, sub=0.04, insert_mean=600, insert_std=200)
             + sample_inversion_reads(g, 16, 1200, 403))
    write_case("f4.case", g, reads)
    for cfg in F4_CONFIGS:
        preset, inv, paired, zd, opt = cfg
        nm = f4_name(*cfg)
        run_ref("f4", "f4.case", preset, 1, nm + ".f4", inv, paired, zd, nm + ".sam", opt)
        gz(nm + ".f4")
        gz(nm + ".sam")
    gz("f4.case")


def main():
    if not have_ref():
        sys.exit("oracle/_ref/ref_dump missing: run `make -C oracle ref` where /root/reference exists")
    os.chdir(HERE)
    # G1/G2/G3..G8: small multi-contig genome with a planted repeat family
    g = rand_genome(101, [30000, 22000, 9000], repeat_unit=200, repeat_copies=25, repeat_div=0.06)
    reads = (sample_reads(g, 90, 150, 201, sub=0.01) + sample_reads(g, 20, 150, 202, sub=0.06, n_rate=0.01)
             + sample_reads(g, 6, 2500, 203, sub=0.01, ins=0.005, dele=0.005)
             + sample_reads(g, 4, 150, 204, random_frac=1.0) + sample_reads(g, 4, 14, 205) + sample_reads(g, 4, 17, 206))
    write_case("small.case", g, reads)
    run_ref("index", "small.case", "small_ref")
    for ext in ("ann", "amb"):
        if os.path.exists("small_ref." + ext):
            os.remove("small_ref." + ext)
    run_ref("ext", "small.case", "small_ref.ext")
    for preset in ("default", "illumina"):
        run_ref("pipe", "small.case", preset, 1, "small_ref.%s.pipe" % preset)
    run_ref("pipe", "small.case", "default", 7, "small_ref.default.seed7.pipe")
    run_ref("pipe", "small.case", "default+mems", 1, "small_ref.mems.pipe")  # "Seeding Technique" = MEMs (binarySeeding.h:460-537)
    sam_goldens()
    reader_goldens()
    # G7: kswcpp cases (all three flag modes, N bases, narrow bands, int16/int32 boundary)
    cases = rand_ksw_cases(600, 301, max_len=120) + rand_ksw_cases(12, 302, long_frac=1.0)
    write_ksw_cases("ksw.case", cases)
    run_ref("ksw", "ksw.case", "ksw_ref.out")
    ksw_scoring_goldens()
    f4_goldens()
    for f in ("small_ref.ext", "small_ref.default.pipe", "small_ref.illumina.pipe", "small_ref.default.seed7.pipe",
              "small_ref.mems.pipe", "ksw_ref.out", "small.case", "ksw.case", "small_ref.bwt", "small_ref.sa", "small_ref.pac"):
        gz(f)
    print("golden vectors written to", HERE)


if __name__ == "__main__":
    main()
