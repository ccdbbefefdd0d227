"""Phase times of one count through the ordered (tree) and the unordered (super-k-mer) entry points.
Usage: python tools/sk_probe.py [n_bases] [k] [iters] [motif]"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import load_package  # noqa: E402

pkg = load_package()
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1_000_000_000
k = int(sys.argv[2]) if len(sys.argv) > 2 else 31
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 3
motif = int(sys.argv[4]) if len(sys.argv) > 4 else 0
with pkg.Context(0) as ctx:
    d = ctx.synth(0xD2A0003, n, motif_len=motif)
    ctx.set_profiling(True)
    for name, fn in (("tree", ctx.count_kmers), ("superkmer", ctx.count_kmers_unordered)):
        best, ph, summ = 1e9, None, None
        for it in range(iters + 1):
            t0 = time.perf_counter()
            h = fn(d, k)
            dt = time.perf_counter() - t0
            if it > 0 and dt < best:
                best, ph = dt, ctx.last_phase_times()
            if it == iters:
                summ = h.summary()
            h.free()
        print(json.dumps({"engine": name, "n_bases": n, "k": k, "motif": motif, "ms": round(best * 1e3, 3),
                          "gkmers_s": round((n - k + 1) / best / 1e9, 2), "summary": summ,
                          "phases_ms": {a: round(b, 3) for a, b in ph}}), flush=True)
