"""Tree against the record engine behind dnagpu_count_kmers_unordered: wall ms of one count for (n_bases, k) pairs, the record
engine forced (DNAGPU_DEBUG_FORCE_SUPERKMER) -- what the engine choice in count_core is set from.
Usage: python tools/engine_probe.py [k,k,...] [n,n,...]"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import load_package  # noqa: E402

pkg = load_package()
ks = [int(x) for x in sys.argv[1].split(",")] if len(sys.argv) > 1 else [21, 22, 23]
ns = [int(float(x)) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else [50_000_000, 100_000_000, 250_000_000, 1_000_000_000, 3_000_000_000]
with pkg.Context(0) as ctx:
    for n in ns:
        d = ctx.synth(0xD2A0001, n)
        for k in ks:
            out = {"n_bases": n, "k": k}
            for name, flags in (("tree_ms", 0), ("records_ms", pkg.DEBUG_FORCE_SUPERKMER)):
                ctx.set_debug(flags)
                fn = ctx.count_kmers_unordered if flags else ctx.count_kmers
                best = 1e9
                for it in range(4):
                    ctx.synchronize()
                    t0 = time.perf_counter()
                    h = fn(d, k)
                    ctx.synchronize()
                    best = min(best, time.perf_counter() - t0)
                    dist = h.distinct
                    h.free()
                out[name] = round(best * 1e3, 3)
                out["distinct_" + name[:4]] = dist
            ctx.set_debug(0)
            print(json.dumps(out), flush=True)
        d.free()
        ctx.trim()
