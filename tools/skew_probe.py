"""Count time on adversarial / low-complexity inputs (one GPU): poly-A, dinucleotide repeat, a short tandem
repeat, half motif / half random, and random with one planted heavy hitter region.
Usage: skew_probe.py [n_bases] [k]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import load_package  # noqa: E402

pkg = load_package()
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1_000_000_000
k = int(sys.argv[2]) if len(sys.argv) > 2 else 31
nw = (n + 31) // 32


def tail_zero(w):
    r = n % 32
    if r:
        w[-1] &= np.uint64((1 << (2 * r)) - 1)
    return w


def rnd(seed):
    rng = np.random.default_rng(seed)
    return rng.integers(0, 2**63, nw, dtype=np.uint64) * np.uint64(2) + rng.integers(0, 2, nw, dtype=np.uint64)


def cases():
    yield "poly-A", np.zeros(nw, dtype=np.uint64)
    yield "(AT)n", np.full(nw, 0x4444444444444444, dtype=np.uint64)
    w = rnd(1)
    period = 7                                  # words: a 224-base tandem repeat
    yield "224-base tandem repeat", np.tile(w[:period], nw // period + 1)[:nw].copy()
    w2 = rnd(2)
    w2[: nw // 2] = 0                           # half poly-A, half random
    yield "half poly-A half random", w2
    w3 = rnd(3)
    w3[: nw // 100] = np.uint64(0x4444444444444444)   # 1 % heavy hitters
    yield "1% (AT)n + random", w3
    yield "random", rnd(4)
    for copies in (8_000, 20_000, 100_000):   # a 64-base motif planted `copies` times: 34 k-mers just over a leaf
        w5 = rnd(5)
        step = (nw // copies) & ~1
        w5[0:step * copies:step] = np.uint64(0x1234567890ABCDEF)
        w5[1:step * copies:step] = np.uint64(0x0FEDCBA987654321)
        yield f"random + {copies} x 64-base motif", w5


with pkg.Context(0) as ctx:
    ctx.set_profiling(True)
    for name, words in cases():
        d = ctx.upload(tail_zero(words), n)
        best, ph = 1e9, None
        for _ in range(2):
            t0 = time.perf_counter()
            h = ctx.count_kmers(d, k)
            dt = time.perf_counter() - t0
            if dt < best:
                best, ph = dt, ctx.last_phase_times()
            tot, dist = h.total, h.distinct
            h.free()
        print(f"{name:28s} {best*1e3:9.2f} ms  rows {tot} groups {dist}  phases {len(ph)}",
              flush=True)
        if os.environ.get("PHASES"):
            print("    ", [(a, round(b, 2)) for a, b in ph], flush=True)
        d.free()
        ctx.trim()
