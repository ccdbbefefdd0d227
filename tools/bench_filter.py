"""Config-5 shaped timing of the fused filter (device outputs): per pattern, wall time per call and
the two kernels' device times (HIP events inside the library).
Usage: python tools/bench_filter.py [n_bases] [iters]"""
import ctypes as C
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import load_package  # noqa: E402

pkg = load_package()
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 100_000_000
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 20
k = 21
nk = n - k + 1
with pkg.Context(0) as ctx:
    d = ctx.synth(0xD2A0001, n)
    kb = ctx.buffer_alloc(nk * 8)
    pb = ctx.buffer_alloc(nk * 8)
    for name, flt in (("contains NNNNNNNNNNWSNNNNNNNNN (1/4)", pkg.Filter.contains("NNNNNNNNNNWSNNNNNNNNN")),
                      ("contains RYNNNNNNNNNNNNNNNNNNN (1/4)", pkg.Filter.contains("RYNNNNNNNNNNNNNNNNNNN")),
                      ("starts_with ACG (1/64)", pkg.Filter.starts_with(3, 0b111000)),
                      ("contains ACGNNNNNNNNNNNNNNNNNN (1/64)", pkg.Filter.contains("ACG" + "N" * 18)),
                      ("contains ACNNNNNNNNNNNNNNNNGTN (1/256)", pkg.Filter.contains("ACNNNNNNNNNNNNNNNNGTN")),
                      ("equals (k singletons)", pkg.Filter.equals(21, 0x123456789ab & ((1 << 42) - 1))),
                      ("contains 21 x N (all rows)", pkg.Filter.contains("N" * 21))):
        for _ in range(3):
            m = ctx.count_matches_device(d, k, flt, 0, nk, C.c_void_p(kb), C.c_void_p(pb), nk)
        ctx.synchronize()
        t0 = time.perf_counter()
        for _ in range(iters):
            m = ctx.count_matches_device(d, k, flt, 0, nk, C.c_void_p(kb), C.c_void_p(pb), nk)
        dt = (time.perf_counter() - t0) / iters
        ctx.set_profiling(True)
        m = ctx.count_matches_device(d, k, flt, 0, nk, C.c_void_p(kb), C.c_void_p(pb), nk)
        ph = dict(ctx.last_phase_times())
        ctx.set_profiling(False)
        print(json.dumps({"filter": name, "n_bases": n, "matches": m, "ms_per_call": round(dt * 1e3, 4),
                          "grows_per_s": round(nk / dt / 1e9, 1),
                          "alg_GBps": round((n / 4 + 16 * m) / dt / 1e9, 1),
                          "kernel_ms": {a: round(b, 4) for a, b in ph.items()}}), flush=True)
