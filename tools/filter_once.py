"""One fused-filter workload (config 5 shape) for profilers: python tools/filter_once.py [n_bases] [pattern] [iters]"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import load_package  # noqa: E402

pkg = load_package()
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 100_000_000
pat = sys.argv[2] if len(sys.argv) > 2 else "NNNNNNNNNNWSNNNNNNNNN"
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 3
k = len(pat)
nk = n - k + 1
with pkg.Context(0) as ctx:
    d = ctx.synth(0xD2A0001, n)
    kb, pb = ctx.buffer_alloc(nk * 8), ctx.buffer_alloc(nk * 8)
    flt = pkg.Filter.contains(pat)
    for _ in range(iters):
        m = ctx.count_matches_device(d, k, flt, 0, nk, C.c_void_p(kb), C.c_void_p(pb), nk)
    print("matches", m)
