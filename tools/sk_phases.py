"""Phase times of the record engine (forced) for several k at one size: python tools/sk_phases.py n_bases k[,k...] [iters]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import load_package  # noqa: E402

pkg = load_package()
n = int(float(sys.argv[1]))
ks = [int(x) for x in sys.argv[2].split(",")]
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 5
with pkg.Context(0) as ctx:
    ctx.set_profiling(True)
    ctx.set_debug(pkg.DEBUG_FORCE_SUPERKMER)
    d = ctx.synth(0xD2A0001, n)
    for k in ks:
        best = {}
        for _ in range(iters):
            h = ctx.count_kmers_unordered(d, k)
            for a, b in ctx.last_phase_times():
                best[a] = min(best.get(a, 1e9), b)
            dist = h.distinct
            h.free()
        print(k, dist, "sum %.3f" % sum(best.values()), {a: round(b, 3) for a, b in best.items() if b > 0.02}, flush=True)
