"""Round 4: the window minimum carries its position -- order = hash bits 7..31 of sk_mix_raw (two v_mul_u32_u24), leftmost among
equals; records are cut where hash OR position changes; d0 comes from the hash, d1 / d2 from the minimum m-mer's value.  Compares records per k-mer and bucket evenness
with round 3's scheme (full 32-bit order, digits from hash bits 0..23): tools/hash_eval2.py [N]"""
import sys
import numpy as np
from numpy.lib.stride_tricks import sliding_window_view

rng = np.random.default_rng(1)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 20_000_000
W = 17
bases = rng.integers(0, 4, size=N + 14, dtype=np.uint32)
v = np.zeros(N, dtype=np.uint32)
for i in range(15):
    v |= bases[i:i + N] << np.uint32(2 * i)


def mix(h):
    h = h.copy(); h += h << np.uint32(11); h ^= h >> np.uint32(7); h += h << np.uint32(17); return h


def digits(word24, c0):
    g = ((word24 & np.uint32(0xFFFFFF)).astype(np.uint64) * np.uint64(0x9E3779) & np.uint64(0xFFFFFFFF)).astype(np.uint32)
    d0 = ((g >> np.uint32(16)).astype(np.uint64) * c0 >> np.uint64(16)).astype(np.uint32)
    return d0, (g >> np.uint32(6)) & np.uint32(511), (g >> np.uint32(2)) & np.uint32(15)


def report(name, lens, d0, d1, d2, n):
    nb = max(16, n // 2700)
    b1 = max(1, int(np.ceil(np.log2(max(1, nb / 68 / 16)))))
    idx = (d0.astype(np.int64) * (1 << b1) + (d1 & np.uint32((1 << b1) - 1))) * 16 + d2
    cnt = np.bincount(idx, weights=lens, minlength=68 * (1 << b1) * 16)
    c0cnt = np.bincount(d0, weights=lens, minlength=68)
    print(f"{name:28s} k-mers/record {n / len(lens):.3f} longest {lens.max()}  buckets {len(cnt)} mean {cnt.mean():.0f} "
          f"cv {cnt.std() / cnt.mean():.3f} max/mean {cnt.max() / cnt.mean():.2f} frac>4096 {np.mean(cnt > 4096):.4f} | "
          f"coarse max/mean {c0cnt.max() / c0cnt.mean():.4f} min/mean {c0cnt.min() / c0cnt.mean():.4f}")


h = mix(v)
# round 3: min of the full hash, records = runs of equal hash
hm = sliding_window_view(h, W).min(axis=1)
n = len(hm)
brk = np.flatnonzero(hm[1:] != hm[:-1])
starts = np.concatenate([[0], brk + 1]); ends = np.concatenate([brk, [n - 1]])
report("round 3 (32-bit order)", ends - starts + 1, *digits(hm[starts], np.uint64(68)), n)
# round 4 as built: sk_mix_raw = two 24-bit multiplies, order = hash bits 7..31 then position (leftmost among equals: the
# position as the low part of a 64-bit key makes that the plain minimum); d0 from the hash, d1 / d2 from the m-mer's VALUE
def mix4(v):
    lo = (v & np.uint32(0xFFFFFF)).astype(np.uint64) * np.uint64(0x9E3779)
    hi = (v >> np.uint32(24)).astype(np.uint64) * np.uint64(0x85EBCB)
    return ((lo + hi) & np.uint64(0xFFFFFFFF)).astype(np.uint32)


def digits4(hmin25, v, c0):
    g = ((hmin25 & np.uint32(0xFFFFFF)).astype(np.uint64) * np.uint64(0x9E3779) & np.uint64(0xFFFFFFFF)).astype(np.uint32)
    d0 = ((g >> np.uint32(16)).astype(np.uint64) * c0 >> np.uint64(16)).astype(np.uint32)
    f = (v.astype(np.uint64) * np.uint64(0x9E3779B1) & np.uint64(0xFFFFFFFF)).astype(np.uint32)
    return d0, (f >> np.uint32(23)) & np.uint32(511), (f >> np.uint32(19)) & np.uint32(15)


h4 = mix4(v)
key = (h4 >> np.uint32(7)).astype(np.uint64) << np.uint64(32) | np.arange(len(h4), dtype=np.uint64)
km = sliding_window_view(key, W).min(axis=1)
brk = np.flatnonzero(km[1:] != km[:-1])
starts = np.concatenate([[0], brk + 1]); ends = np.concatenate([brk, [n - 1]])
g = (km[starts] >> np.uint64(32)).astype(np.uint32)
vmin = v[(km[starts] & np.uint64(0xFFFFFFFF)).astype(np.int64)]
report("round 4 (25 bits + position)", ends - starts + 1, *digits4(g, vmin, np.uint64(68)), n)
ties = np.count_nonzero((km[1:] >> np.uint64(32)) == (km[:-1] >> np.uint64(32))) - np.count_nonzero(km[1:] == km[:-1])
print(f"rows whose minimum keeps its hash but moves (ties of 25-bit hashes / repeated m-mers): {ties} of {n} ({ties / n:.2e})")
