"""Randomised parity soak of the unordered count (super-k-mer engine, all its paths) against the oracle: random lengths,
k in 20..32, repeat motifs, planted heavy stretches, windows, and the debug flags that steer the paths (engine forced;
heavy mid buckets expanded instead of split; level 1 speculative / exact / forced fall-back).  Usage: python tools/fuzz_unordered.py [cases] [max_n] [seed]"""
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
rng = np.random.default_rng(int(sys.argv[3]) if len(sys.argv) > 3 else 4321)
t_start = time.time()
bad = 0
with pkg.Context(0) as ctx:
    for c in range(cases):
        n = int(rng.integers(64, max_n)) if rng.random() < 0.8 else int(rng.integers(64, 50_000))
        k = int(rng.integers(20, 33))
        seed = int(rng.integers(0, 2**31))
        motif = int(rng.choice([0, 0, 1, 2, 3, 7, 31, 64, 1000, 50_000]))
        if motif and motif * 2 < n:
            words = orc.synth_words_repeat(seed, n, motif)
        else:
            motif = 0
            words = orc.synth_words(seed, n)
        # near-copies: the second half repeats the first with point changes every ~10 / 30 / 100 / 300 bases -- records
        # that share their minimizer and agree over PART of their k-mers (sk_count's record-against-record test: the
        # boundary "bases agree over at least k - m around the m-mer")
        mutated = 0
        if rng.random() < 0.3 and len(words) > 8:
            mutated = int(rng.choice([10, 30, 100, 300]))
            words = words.copy()
            nw = len(words)
            half = nw // 2
            shift_words = int(rng.integers(0, 3))       # (the copy need not start on the same word phase ...)
            words[half:half + half - shift_words] = words[shift_words:half]
            n_mut = max(1, (half * 32) // mutated)
            pos = rng.integers(half * 32, nw * 32, n_mut)
            for pb in pos:
                wi, sh = int(pb) >> 5, 2 * (int(pb) & 31)
                words[wi] ^= np.uint64(int(rng.integers(1, 4)) << sh)
            if rng.random() < 0.5:                      # (... nor on the same base phase: one base inserted at the half)
                carry = np.uint64(int(rng.integers(0, 4)))
                for wi in range(half, nw):
                    nxt = words[wi] >> np.uint64(62)
                    words[wi] = (words[wi] << np.uint64(2)) | carry
                    carry = nxt
            r = n % 32
            if r:
                words[-1] &= np.uint64((1 << (2 * r)) - 1)
        planted = 0
        if rng.random() < 0.4 and len(words) > 8:
            words = words.copy()
            nw = len(words)
            for _ in range(int(rng.integers(1, 4))):
                planted += 1
                val = np.uint64(rng.choice([0, 0xFFFFFFFFFFFFFFFF, 0x4444444444444444, 0x0000000100000001, int(rng.integers(0, 2**63))]))
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
        count = int(rng.integers(1, nk - first + 1)) if nk and first else nk
        if count == 0:
            d.free()
            continue
        keys = orc.generate_kmers(words, n, k, first, count, faithful=False)
        ok, oc = orc.count_keys(keys)
        flags = int(rng.choice([pkg.DEBUG_FORCE_SUPERKMER, pkg.DEBUG_FORCE_SUPERKMER, pkg.DEBUG_FORCE_SUPERKMER | pkg.DEBUG_HEAVY_EXPAND]))
        # level 1: speculative regions (the default; repeats overflow them and fall back), the exact level, the forced fall-back
        flags |= int(rng.choice([0, 0, pkg.DEBUG_NO_SPEC1, pkg.DEBUG_SPEC1_OVERFLOW]))
        # level 0: slabs from a sampled histogram (forced: the sequences here are short), the exact pair, the forced fall-back
        flags |= int(rng.choice([pkg.DEBUG_SLAB0, pkg.DEBUG_SLAB0, pkg.DEBUG_NO_SLAB0, pkg.DEBUG_SLAB0_OVERFLOW]))
        ctx.set_debug(flags)
        try:
            h = ctx.count_kmers_unordered(d, k, first, count)
        finally:
            ctx.set_debug(0)
        gk, gc = h.download()
        order = np.argsort(gk, kind="stable")
        # (with DEBUG_HEAVY_EXPAND a sequence that is mostly repeats is counted by the ordered engine: both orders are fine)
        good = h.distinct == len(ok) and np.array_equal(gk[order], ok) and np.array_equal(gc[order], oc) and \
            h.summary() == orc.hist_summary(ok, oc) and h.total == count
        h.free()
        d.free()
        if not good:
            bad += 1
            print(f"MISMATCH case {c}: n={n} k={k} seed={seed} motif={motif} mutated={mutated} planted={planted} first={first} count={count} flags={flags}",
                  flush=True)
        if c % 25 == 24:
            print(f"... {c + 1} cases, {bad} mismatches, {time.time() - t_start:.0f} s", flush=True)
print(f"{cases} cases, {bad} mismatches, {time.time() - t_start:.0f} s")
sys.exit(1 if bad else 0)
