"""Level 0 of the record engine without its histogram sweep (slabs sized from a sampled histogram) against the exact pair,
on one GPU: for every (n_bases, k, motif) the summary of dnagpu_count_kmers_unordered with the slab sweep
(DNAGPU_DEBUG_SLAB0), with the exact pair (DNAGPU_DEBUG_NO_SLAB0) and with DNAGPU_DEBUG_SLAB0_OVERFLOW (slab sweep, then
the fall-back), the phase times of each, and (n <= 300 M) the ordered engine's summary beside them.
Usage: python tools/slab0_probe.py [case ...]   case = n_bases:k[:motif]"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import load_package  # noqa: E402

pkg = load_package()
cases = sys.argv[1:] or ["3e9:31", "3e9:31:1000", "248956422:31", "1e8:21", "4e7:27:64", "1000003:31", "5000:25"]
modes = [("slab", pkg.DEBUG_SLAB0), ("exact", pkg.DEBUG_NO_SLAB0), ("overflow", pkg.DEBUG_SLAB0_OVERFLOW)]
bad = 0
with pkg.Context(0) as ctx:
    ctx.set_profiling(True)
    for case in cases:
        f = case.split(":")
        n, k, motif = int(float(f[0])), int(f[1]), int(f[2]) if len(f) > 2 else 0
        d = ctx.synth(0xD2A0003, n, motif)
        ref = None
        if n <= 300_000_000:
            ctx.set_debug(0)
            h = ctx.count_kmers(d, k)
            ref = h.summary()
            h.free()
        for name, flag in modes:
            ctx.set_debug(flag | pkg.DEBUG_FORCE_SUPERKMER)
            best, wall, s = {}, 1e9, None
            for it in range(4):
                ctx.synchronize()
                t0 = time.perf_counter()
                h = ctx.count_kmers_unordered(d, k)
                wall = min(wall, (time.perf_counter() - t0) * 1e3)
                for a, b in ctx.last_phase_times():
                    best[a] = min(best.get(a, 1e9), b)
                if it == 0:
                    s = h.summary()
                h.free()
            if ref is None:
                ref = s
            ok = s == ref
            bad += 0 if ok else 1
            l1 = {a: round(b, 3) for a, b in best.items() if a in ("sk_plan0", "sk_sample0", "sk_hist0", "sk_prefix0", "sk_scatter0", "sk_spec1", "sk_hist1", "sk_scatter1")}
            print(json.dumps({"n_bases": n, "k": k, "motif": motif, "mode": name, "ok": ok, "wall_ms": round(wall, 3),
                              "levels01_ms": round(sum(l1.values()), 3), "levels01": l1, "summary": list(s)}), flush=True)
        ctx.set_debug(0)
        d.free()
print("MISMATCHES", bad, flush=True)
sys.exit(1 if bad else 0)
