"""Fixed cost of one count call: wall time of the unordered / ordered count and of the two record halves over inputs small
enough that the kernels' own work is negligible (what is left is host synchronisations, small launches, allocations).
Usage: python tools/overhead_probe.py"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import load_package  # noqa: E402

pkg = load_package()
k = 31
with pkg.Context(0) as ctx:
    for n in (100_000, 1_000_000, 10_000_000, 50_000_000):
        d = ctx.synth(7, n)
        rows = n - k + 1
        out = {"n_bases": n}
        for name, fn in (("tree", lambda: ctx.count_kmers(d, k)), ("unordered_auto", lambda: ctx.count_kmers_unordered(d, k))):
            best = 1e9
            for it in range(6):
                ctx.synchronize()
                t0 = time.perf_counter()
                h = fn()
                ctx.synchronize()
                best = min(best, time.perf_counter() - t0)
                h.free()
            out[name + "_ms"] = round(best * 1e3, 3)
        ctx.set_debug(pkg.DEBUG_FORCE_SUPERKMER)
        best = b0 = b1 = 1e9
        for it in range(6):
            ctx.synchronize()
            t0 = time.perf_counter()
            h = ctx.count_kmers_unordered(d, k)
            ctx.synchronize()
            best = min(best, time.perf_counter() - t0)
            h.free()
            t0 = time.perf_counter()
            r = ctx.sk_records(d, k, 0, rows, rows)
            ctx.synchronize()
            t1 = time.perf_counter()
            nb = len(r.offsets) - 1
            pieces = [(r.device_ptr + 16 * int(r.offsets[b]), int(r.offsets[b + 1] - r.offsets[b]), b) for b in range(nb)]
            t2 = time.perf_counter()
            h = ctx.count_records(pieces, k, rows)
            ctx.synchronize()
            t3 = time.perf_counter()
            b0, b1 = min(b0, t1 - t0), min(b1, t3 - t2)
            h.free()
            r.free()
        ctx.set_debug(0)
        out["superkmer_ms"] = round(best * 1e3, 3)
        out["sk_records_ms"] = round(b0 * 1e3, 3)
        out["count_records_ms"] = round(b1 * 1e3, 3)
        print(json.dumps(out), flush=True)
        d.free()
