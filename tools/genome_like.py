"""A genome-LIKE synthetic sequence (what the uniform configs are not): random bases with
   * an Alu-like family -- copies of one 300-base consensus every ~3 kb (10 % of the bases), each diverged by 1 - 15 %,
   * exact segmental duplications -- 10 kb pieces copied elsewhere (2 %),
   * microsatellites and homopolymer runs of 20 - 60 bases (1 %).
packed_words(n_bases, seed) -> uint64 words in the reference's layout (A=00 T=01 C=10 G=11, LSB first).  numpy, host side;
bench.py --genome-like uploads it.  Usage as a script: python tools/genome_like.py [n_bases] -> statistics."""
import sys

import numpy as np


def codes(n, seed=0x6E0):
    rng = np.random.default_rng(seed)
    c = rng.integers(0, 4, n, dtype=np.uint8)
    # Alu-like copies
    cons = rng.integers(0, 4, 300, dtype=np.uint8)
    n_alu = n // 3000
    pos = rng.integers(0, max(n - 300, 1), n_alu)
    div = rng.uniform(0.01, 0.15, n_alu)
    for p, d in zip(pos, div):
        copy = cons.copy()
        m = rng.random(300) < d
        copy[m] = rng.integers(0, 4, int(m.sum()), dtype=np.uint8)
        c[p:p + 300] = copy[:max(0, min(300, n - p))]
    # exact segmental duplications
    seg = 10_000
    for _ in range(max(1, (n // 50) // seg)):
        a, b = rng.integers(0, max(n - seg, 1), 2)
        c[b:b + seg] = c[a:a + seg][:max(0, min(seg, n - b))]
    # microsatellites / homopolymers
    units = [np.array(u, dtype=np.uint8) for u in ([0], [1], [0, 2], [0, 1], [2, 0, 3], [1, 1, 0, 3])]
    n_ms = (n // 100) // 40
    for p in rng.integers(0, max(n - 64, 1), n_ms):
        u = units[int(rng.integers(0, len(units)))]
        ln = int(rng.integers(20, 61))
        c[p:p + ln] = np.resize(u, ln)[:max(0, min(ln, n - p))]
    return c


def packed_words(n, seed=0x6E0):
    c = codes(n, seed).astype(np.uint64)
    nw = (n + 31) // 32
    pad = np.zeros(nw * 32, dtype=np.uint64)
    pad[:n] = c
    pad = pad.reshape(nw, 32)
    sh = (np.arange(32, dtype=np.uint64) * np.uint64(2))
    return (pad << sh).sum(axis=1, dtype=np.uint64)


if __name__ == "__main__":
    n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 10_000_000
    w = packed_words(n)
    print(n, "bases ->", len(w), "words")
