"""Helper of test_gpu_round2.py::test_dp_scratch_budget_paths: aligns a fixed read set and prints a digest of every
alignment.  Run in its own process because the DP scratch budget (MA_KSW_SCRATCH_MB) is read once per process."""
import hashlib
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from ma_testlib import rand_genome, sample_reads  # noqa: E402


def main():
    import ma_amd
    g = rand_genome(29, [1500000, 700000], repeat_unit=300, repeat_copies=80, repeat_div=0.08)
    reads = (sample_reads(g, 300, 150, 61, sub=0.01) + sample_reads(g, 6, 20000, 62, sub=0.03, ins=0.03, dele=0.04)
             + sample_reads(g, 12, 4000, 63, sub=0.005, ins=0.003, dele=0.003) + sample_reads(g, 2, 45000, 64, sub=0.02, ins=0.01, dele=0.01))
    idx = ma_amd.Index.build(g)
    b = ma_amd.Batch(idx, ma_amd.Params.preset("default"), len(reads), sum(len(r) for r in reads) + 64)
    b.set_reads(reads)
    b.align()
    b.sync()
    h = hashlib.sha256()
    for a in b.alignments() + b.mapq_alignments():
        h.update(np.ascontiguousarray(a).tobytes())
    print("DIGEST", h.hexdigest(), len(b.alignments()[1]))


if __name__ == "__main__":
    main()
