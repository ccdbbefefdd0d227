"""Runs `iters` counts of k over n synthetic bases (for rocprofv3 runs). Usage: count_once.py n k iters [motif]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import load_package  # noqa: E402

pkg = load_package()
n, k, iters = int(float(sys.argv[1])), int(sys.argv[2]), int(sys.argv[3])
motif = int(sys.argv[4]) if len(sys.argv) > 4 else 0
with pkg.Context(0) as ctx:
    ctx.set_profiling(True)
    d = ctx.synth(0xD2A0003, n, motif)
    best = {}
    wall = 1e9
    for _ in range(iters):
        t0 = time.perf_counter()
        h = ctx.count_kmers(d, k)
        wall = min(wall, (time.perf_counter() - t0) * 1e3)
        ph = ctx.last_phase_times()
        if os.environ.get("ALL_PHASES") and _ == iters - 1:
            print([(a, round(b, 2)) for a, b in ph], flush=True)
        for a, b in ph:
            best[a] = min(best.get(a, 1e9), b)
        distinct = h.distinct
        h.free()
    print(distinct, "min over", iters, [(a, round(b, 3)) for a, b in best.items()], "sum", round(sum(best.values()), 2), "wall ms", round(wall, 2), flush=True)
