"""Per-call latency of the boundary at small sizes (one PostgreSQL row = one call): count_kmers, generate_kmers
(keys to host) and a filtered extraction, input resident.  Usage: latency_probe.py [k]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import load_package  # noqa: E402

pkg = load_package()
k = int(sys.argv[1]) if len(sys.argv) > 1 else 21
with pkg.Context(0) as ctx:
    flt = pkg.Filter.contains("N" * (k - 2) + "WS")
    for n in (100, 1_000, 10_000, 100_000, 1_000_000, 10_000_000):
        d = ctx.synth(7, n)
        res = {}
        for name, fn in (("count", lambda: ctx.count_kmers(d, k).free()),
                         ("generate->host", lambda: ctx.generate_kmers(d, k)),
                         ("filter->host", lambda: ctx.generate_kmers_filtered(d, k, flt))):
            for _ in range(3):
                fn()
            reps = 20
            t0 = time.perf_counter()
            for _ in range(reps):
                fn()
            res[name] = (time.perf_counter() - t0) / reps * 1e6
        print(f"n={n:>9}: " + "  ".join(f"{a} {b:9.1f} us" for a, b in res.items()), flush=True)
        d.free()
