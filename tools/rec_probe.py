"""Records of the super-k-mer engine's level 0 on synthetic sequence: how many, how long, how many SK_REC_MULTI, how the coarse
buckets fill (dnagpu_sk_records + a download of the records).  Usage: python tools/rec_probe.py [n_bases] [k]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import load_package  # noqa: E402

pkg = load_package()
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 100_000_000
k = int(sys.argv[2]) if len(sys.argv) > 2 else 31
with pkg.Context(0) as ctx:
    d = ctx.synth(0xD2A0003, n)
    rows = n - k + 1
    r = ctx.sk_records(d, k, 0, rows, rows)
    off = r.offsets
    n_rec = r.n_records
    recs = ctx.download_u64(r.device_ptr, 2 * n_rec).reshape(-1, 2)
    y = recs[:, 1]
    lens = ((y >> np.uint64(44)) & np.uint64(31)).astype(np.int64) + 1
    multi = (y >> np.uint64(58)) & np.uint64(1)
    pos = (y >> np.uint64(39)) & np.uint64(31)
    null = y >> np.uint64(63)
    print(f"rows {rows}  records {n_rec}  k-mers/record {lens.sum() / n_rec:.3f}  (k-mers in records {int(lens.sum())})")
    print(f"NULL {int(null.sum())}  multi {int(multi.sum())}  len histogram {np.bincount(lens)[:24]}")
    print(f"quads per record {np.mean((lens + 3) // 4):.3f}; minimizer offset histogram (non-multi) {np.bincount(pos[multi == 0].astype(np.int64), minlength=18)}")
    sizes = np.diff(off)
    print(f"coarse buckets {len(sizes)}: records min {sizes.min()} mean {sizes.mean():.0f} max {sizes.max()}")
    bad = np.count_nonzero((multi == 0) & ((pos.astype(np.int64) < lens - 1) | (pos.astype(np.int64) > k - (15 if k >= 23 else 13))))
    print(f"non-multi records whose offset is outside [len - 1, w - 1]: {bad}")
    r.free()
    d.free()
