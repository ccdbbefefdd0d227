#!/usr/bin/env python3
"""tools/make_digests.py -- the CPU oracle's (total, distinct, unique, checksum) of the BASELINE.json count configs at
their FULL sizes, written to tests/golden/config_digests.json.

Run in the build container (CPU only, no GPU, nothing from /root/reference at run time):

    python tools/make_digests.py [--only 2 3 4 3m1000 4m1000] [--workers 5] [--slices 32]

The digests are what test_config{2,3,4}_* assert BOTH GPU engines' dnagpu_hist_summary against, so parity at 248,956,422
and 3,000,000,000 bases is oracle-vs-GPU, not engine-vs-engine.  The oracle cannot hold 3 G keys at once (24 GB of keys
plus a 100 GB hash table), so the key space is cut into slices (oracle/kmer_oracle.c: orc_count_kmers_slice -- every copy
of a k-mer lands in the same slice); each slice is one sweep over the synthetic stream + the oracle's hash aggregate,
and total / distinct / unique / checksum add up over the slices (the checksum is a wrapping sum over groups).
TEST INFRASTRUCTURE: writes a fixture, is never imported by the product.
"""
import argparse
import json
import os
import sys
import time
from multiprocessing import Pool

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

# seeds and sizes of the configs; "m1000" = the repeat-rich variant (a 1000-base motif tiled over
# the second half of the sequence)
CASES = {
    "2": {"n_bases": 100_000_000, "k": 21, "seed": 0xD2A0001, "motif": 0},
    "3": {"n_bases": 248_956_422, "k": 31, "seed": 0xD2A0002, "motif": 0},
    "4": {"n_bases": 3_000_000_000, "k": 31, "seed": 0xD2A0003, "motif": 0},
    "2m1000": {"n_bases": 100_000_000, "k": 21, "seed": 0xD2A0001, "motif": 1000},
    "3m1000": {"n_bases": 248_956_422, "k": 31, "seed": 0xD2A0002, "motif": 1000},
    "4m1000": {"n_bases": 3_000_000_000, "k": 31, "seed": 0xD2A0003, "motif": 1000},
}
OUT = os.path.join(ROOT, "tests", "golden", "config_digests.json")
M64 = (1 << 64) - 1

_words = {}


def _slice(job):
    import oracle as orc
    name, s, n_slices = job
    c = CASES[name]
    if name not in _words:
        _words.clear()
        _words[name] = (orc.synth_words_repeat(c["seed"], c["n_bases"], c["motif"]) if c["motif"]
                        else orc.synth_words(c["seed"], c["n_bases"]))
    t0 = time.time()
    keys, counts = orc.count_kmers_slice(_words[name], c["n_bases"], c["k"], s, n_slices)
    total, distinct, unique, checksum = orc.hist_summary(keys, counts)
    return name, s, total, distinct, unique, checksum, int(counts.max()) if len(counts) else 0, time.time() - t0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", nargs="*", default=["2", "3", "4"])
    ap.add_argument("--workers", type=int, default=5)
    ap.add_argument("--slices", type=int, default=0, help="0: one slice per 100 M k-mers")
    args = ap.parse_args()
    try:
        with open(OUT) as f:
            out = json.load(f)
    except Exception:
        out = {}
    for name in args.only:
        c = CASES[name]
        n_slices = args.slices or max(1, -(-c["n_bases"] // 100_000_000))
        t0 = time.time()
        tot = [0, 0, 0, 0]
        max_count = 0
        with Pool(min(args.workers, n_slices)) as pool:
            for r in pool.imap_unordered(_slice, [(name, s, n_slices) for s in range(n_slices)]):
                tot = [(a + b) & M64 for a, b in zip(tot, r[2:6])]
                max_count = max(max_count, r[6])
                print(f"  {name}: slice {r[1]}/{n_slices}: {r[3]} groups, {r[7]:.0f} s", flush=True)
        assert tot[0] == c["n_bases"] - c["k"] + 1, (tot, c)
        out[name] = {"n_bases": c["n_bases"], "k": c["k"], "seed": c["seed"], "motif": c["motif"],
                     "total": tot[0], "distinct": tot[1], "unique": tot[2], "checksum": tot[3],
                     "max_count": max_count,
                     "made_by": f"tools/make_digests.py: CPU oracle, {n_slices} key-space slices (orc_count_kmers_slice)"}
        print(name, out[name], f"{time.time() - t0:.0f} s", flush=True)
        with open(OUT, "w") as f:
            json.dump(out, f, indent=1, sort_keys=True)
            f.write("\n")


if __name__ == "__main__":
    main()
