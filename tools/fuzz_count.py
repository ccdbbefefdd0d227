"""Randomised parity soak of the count (and of generate_kmers / the owner partition) against the oracle:
random lengths, k, seeds, repeat motifs and windows.  Usage: python tools/fuzz_count.py [cases] [max_n] [seed]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import oracle as orc  # noqa: E402
from __graft_entry__ import load_package  # noqa: E402

pkg = load_package()
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
max_n = int(float(sys.argv[2])) if len(sys.argv) > 2 else 3_000_000
rng = np.random.default_rng(int(sys.argv[3]) if len(sys.argv) > 3 else 12345)
t_start = time.time()
bad = 0
with pkg.Context(0) as ctx:
    for c in range(cases):
        n = int(rng.integers(1, max_n)) if rng.random() < 0.8 else int(rng.integers(1, 20_000))
        k = int(rng.integers(1, 33))
        seed = int(rng.integers(0, 2**31))
        motif = int(rng.choice([0, 0, 0, 1, 3, 7, 64, 1000, 50_000]))
        if motif and motif * 2 < n:
            words = orc.synth_words_repeat(seed, n, motif)
        else:
            motif = 0
            words = orc.synth_words(seed, n)
        planted = 0
        if rng.random() < 0.35 and len(words) > 8:        # heavy hitters: constant stretches and strided motifs
            words = words.copy()
            nw = len(words)
            for _ in range(int(rng.integers(1, 4))):
                planted += 1
                val = np.uint64(rng.choice([0, 0x4444444444444444, 0x0000000100000001, int(rng.integers(0, 2**63))]))
                if rng.random() < 0.5:
                    lo = int(rng.integers(0, nw))
                    words[lo:lo + int(rng.integers(1, max(nw // 2, 2)))] = val
                else:
                    step = int(rng.integers(2, max(nw // 50, 3))) & ~1
                    words[0::step] = val
                    words[1::step] = np.uint64(int(val) ^ 0x0FEDCBA987654321)
            r = n % 32
            if r:
                words[-1] &= np.uint64((1 << (2 * r)) - 1)
        d = ctx.upload(words, n)
        nk = max(n - k + 1, 0)
        first = int(rng.integers(0, nk)) if nk and rng.random() < 0.3 else 0
        count = int(rng.integers(0, nk - first + 1)) if nk and first else nk
        keys = orc.generate_kmers(words, n, k, first, count, faithful=False) if count else np.zeros(0, np.uint64)
        ok, oc = orc.count_keys(keys)
        h = ctx.count_kmers(d, k, first, count) if count else None
        good = True
        if h is not None:
            gk, gc = h.download()
            good = np.array_equal(gk, ok) and np.array_equal(gc, oc) and h.summary() == orc.hist_summary(ok, oc)
            h.free()
            W = int(rng.choice([2, 3, 4, 8]))
            if rng.random() < 0.25 and first == 0 and count == nk:
                parts = []
                for o in range(W):
                    ho = ctx.count_kmers_owned(d, k, o, W)
                    parts.append(ho.download())
                    ho.free()
                pk = np.concatenate([p[0] for p in parts])
                pc = np.concatenate([p[1] for p in parts])
                good = good and np.array_equal(pk, ok) and np.array_equal(pc, oc)
        d.free()
        if not good:
            bad += 1
            print(f"MISMATCH case {c}: n={n} k={k} seed={seed} motif={motif} planted={planted} first={first} count={count}", flush=True)
        if c % 20 == 19:
            print(f"{c + 1} cases, {bad} mismatches, {time.time() - t_start:.0f} s", flush=True)
print(f"done: {cases} cases, {bad} mismatches")
sys.exit(1 if bad else 0)
