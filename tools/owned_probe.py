"""Per-rank compute time of the sharded count (without the all-gather) on one GPU:
count_kmers_owned(whole sequence, owner, W) for W in 1, 2, 4, 8.  Usage: owned_probe.py [n_bases] [k]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import load_package  # noqa: E402

pkg = load_package()
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 3_000_000_000
k = int(sys.argv[2]) if len(sys.argv) > 2 else 31
with pkg.Context(0) as ctx:
    ctx.set_profiling(True)
    d = ctx.synth(0xD2A0003, n)
    for W in (1, 2, 4, 8):
        best, bestph = 1e9, None
        for it in range(3):
            t0 = time.perf_counter()
            h = ctx.count_kmers_owned(d, k, W // 2, W)
            dt = time.perf_counter() - t0
            if dt < best:
                best, bestph = dt, ctx.last_phase_times()
            tot, dist = h.total, h.distinct
            h.free()
        print(f"W={W}: {best*1e3:.2f} ms wall, owned rows {tot}, groups {dist}, "
              f"job rate if all ranks take this long: {(n-k+1)/best/1e9:.1f} G k-mers/s", flush=True)
        print("    ", [(a, round(b, 2)) for a, b in bestph], flush=True)
        ctx.trim()
