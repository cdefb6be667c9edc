#!/usr/bin/env python3
"""Pins the index builder at the scale where the reference switches to BWA's bwtLarge construction
(fMIndex.cpp:316-338: doubled text >= 50 Mnt).  Runs HERE (needs /root/reference compiled into oracle/_ref):
  1. restates the device genome generator (ma_amd/csrc/synth.hip k_genome: a pure function of (seed, position)) in numpy,
  2. lets the REFERENCE build pack + FMD-index of that genome (ref_dump index -> vStoreCollection / vStoreFMIndex),
  3. commits SHA-256 of the reference's .bwt / .sa / .pac bytes (hashes only) as tests/golden/large_index.sha256.json.
The -m gpu test builds the same genome with ma_synth_genome_device + ma_index_build_device and compares hashes."""
import hashlib
import json
import os
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "tests"))

SEED = 5
CONTIGS = [12_000_000, 10_500_000, 7_500_001]  # 30 Mnt forward (odd total: exercises the .pac tail), doubled 60 Mnt


def mix64(x):
    x = x + np.uint64(0x9E3779B97F4A7C15)
    x = (x ^ (x >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
    x = (x ^ (x >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
    return x ^ (x >> np.uint64(31))


def h2(seed, a):
    return mix64(mix64(np.array([seed], dtype=np.uint64))[0] ^ a)


def synth_genome(seed, total, repeats=True):
    """k_genome of synth.hip, vectorised."""
    out = np.empty(total, dtype=np.uint8)
    step = 1 << 22
    with np.errstate(over="ignore"):
        for lo in range(0, total, step):
            i = np.arange(lo, min(total, lo + step), dtype=np.uint64)
            b = (h2(seed, i) & np.uint64(3)).astype(np.uint32)
            if repeats:
                blk6, blk3 = i // np.uint64(6000), i // np.uint64(300)
                in6 = h2(seed + 1, blk6) % np.uint64(103) == 0
                in3 = (~in6) & (h2(seed + 4, blk3) % np.uint64(10) == 0)
                for mask, s_unit, s_mut, period, div in ((in6, seed + 2, seed + 3, 6000, 5), (in3, seed + 5, seed + 6, 300, 12)):
                    cb = (h2(s_unit, i % np.uint64(period)) & np.uint64(3)).astype(np.uint32)
                    hm = h2(s_mut, i)
                    mut = (hm % np.uint64(100)) < np.uint64(div)
                    alt = (cb + 1 + ((hm >> np.uint64(8)) % np.uint64(3)).astype(np.uint32)) & 3
                    b = np.where(mask, np.where(mut, alt, cb), b)
            out[lo:lo + len(i)] = b.astype(np.uint8)
    return out


def main():
    from ma_testlib import write_case
    ref = os.path.join(ROOT, "oracle", "_ref", "ref_dump")
    if not os.path.exists(ref):
        sys.exit("oracle/_ref/ref_dump missing: make -C oracle ref")
    total = sum(CONTIGS)
    g = synth_genome(SEED, total)
    contigs, o = [], 0
    for l in CONTIGS:
        contigs.append(g[o:o + l])
        o += l
    with tempfile.TemporaryDirectory(dir="/tmp") as d:
        case = os.path.join(d, "large.case")
        write_case(case, contigs, [])
        env = dict(os.environ, LD_LIBRARY_PATH=os.path.join(ROOT, "oracle", "_ref") + ":" + os.environ.get("LD_LIBRARY_PATH", ""))
        subprocess.check_call([ref, "index", case, os.path.join(d, "ref")], env=env, cwd=d)
        out = {"seed": SEED, "contigs": CONTIGS, "with_repeats": 1,
               "note": "SHA-256 of the files the REFERENCE wrote (FMIndex::vStoreFMIndex, Pack::vStoreCollection) for the genome "
                       "ma_synth_genome_device(seed, sum(contigs), 1) cut into these contigs; doubled text 60 Mnt >= 50 Mnt: "
                       "BWA bwtLarge construction (fMIndex.cpp:316-338)"}
        for ext in ("bwt", "sa", "pac"):
            with open(os.path.join(d, "ref." + ext), "rb") as f:
                data = f.read()
            out[ext] = {"sha256": hashlib.sha256(data).hexdigest(), "bytes": len(data)}
        # a small window of the genome so the numpy generator itself is pinned against the device one
        out["genome_sha256"] = hashlib.sha256(g.tobytes()).hexdigest()
    with open(os.path.join(HERE, "large_index.sha256.json"), "w") as f:
        json.dump(out, f, indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
