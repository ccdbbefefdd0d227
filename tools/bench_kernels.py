"""Throughput of the non-count kernels (generate_kmers, fused filters, kmer_hash) with HIP-event-free
wall timing around synchronous C-ABI calls on device-resident buffers.
Usage: python tools/bench_kernels.py [n_bases]"""
import ctypes as C
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import load_package  # noqa: E402

pkg = load_package()
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 100_000_000
out = {}
with pkg.Context(0) as ctx:
    d = ctx.synth(0xD2A0001, n)
    for k in (21, 31):
        nk = n - k + 1
        buf = ctx.buffer_alloc(nk * 8)
        best = 1e9
        for _ in range(5):
            t0 = time.perf_counter()
            ctx.generate_kmers_device(d, k, 0, nk, buf)
            best = min(best, time.perf_counter() - t0)
        out[f"generate_kmers_k{k}"] = {"ms": best * 1e3, "gkmers_s": nk / best / 1e9,
                                       "GBps_out": nk * 8 / best / 1e9}
        t0 = time.perf_counter()
        import numpy as np
        h = ctx.buffer_alloc(nk * 4)
        best = 1e9
        for _ in range(5):
            t0 = time.perf_counter()
            rc = pkg.lib().dnagpu_kmer_hash(ctx.h, buf, nk, h, 1)
            best = min(best, time.perf_counter() - t0)
        out[f"kmer_hash_k{k}"] = {"ms": best * 1e3, "gkeys_s": nk / best / 1e9, "GBps": nk * 12 / best / 1e9}
        ctx.buffer_free(h)
        ctx.buffer_free(buf)
    k = 21
    nk = n - k + 1
    for name, flt in (("contains_WS(1/4)", pkg.Filter.contains("NNNNNNNNNNWSNNNNNNNNN")),
                      ("contains_RY(1/4)", pkg.Filter.contains("RYNNNNNNNNNNNNNNNNNNN")),
                      ("starts_with_ACG(1/64)", pkg.Filter.starts_with(3, 0b111000)),
                      ("contains_N(all)", pkg.Filter.contains("N" * 21))):
        kb = ctx.buffer_alloc(nk * 8)
        pb = ctx.buffer_alloc(nk * 8)
        best = 1e9
        for _ in range(5):
            t0 = time.perf_counter()
            m = ctx.count_matches_device(d, k, flt, 0, nk, C.c_void_p(kb), C.c_void_p(pb), nk)
            best = min(best, time.perf_counter() - t0)
        out[f"filter_{name}"] = {"ms": best * 1e3, "matches": m, "gkmers_s_scanned": nk / best / 1e9,
                                 "GBps_alg": (n / 4 + 16 * m) / best / 1e9}
        ctx.buffer_free(kb)
        ctx.buffer_free(pb)
for k_, v in out.items():
    print(k_, json.dumps(v))

# text <-> packed (device-resident text in, so the rate is the kernel's, not PCIe's)
with pkg.Context(0) as ctx:
    import numpy as np
    d = ctx.synth(0xD2A0001, n)
    tbuf = ctx.buffer_alloc(n)
    rc = pkg.lib().dnagpu_dna_unpack(ctx.h, d.h, 0, n, tbuf, 1)
    best_u = best_p = 1e9
    for _ in range(5):
        t0 = time.perf_counter()
        pkg.lib().dnagpu_dna_unpack(ctx.h, d.h, 0, n, tbuf, 1)
        best_u = min(best_u, time.perf_counter() - t0)
    for _ in range(5):
        h = C.c_void_p()
        t0 = time.perf_counter()
        rc = pkg.lib().dnagpu_dna_pack(ctx.h, C.cast(tbuf, C.c_char_p), n, 1, C.byref(h), None, None)
        best_p = min(best_p, time.perf_counter() - t0)
        pkg.lib().dnagpu_dna_free(ctx.h, h)
    print("dna_pack", json.dumps({"ms": best_p * 1e3, "gbases_s": n / best_p / 1e9, "GBps": n * 1.25 / best_p / 1e9, "rc": rc}))
    print("dna_unpack", json.dumps({"ms": best_u * 1e3, "gbases_s": n / best_u / 1e9, "GBps": n * 1.25 / best_u / 1e9}))
