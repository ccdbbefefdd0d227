"""Host-boundary costs outside bench.py's `value`: upload of the packed sequence (dnagpu_dna_upload) and
download of result windows (dnagpu_hist_download).  Usage: python tools/pcie_probe.py [n_bases]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import load_package  # noqa: E402

pkg = load_package()
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 3_000_000_000
k = 31
with pkg.Context(0) as ctx:
    d0 = ctx.synth(0xD2A0003, n)
    words = d0.download()
    d0.free()
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        d = ctx.upload(words, n)
        best = min(best, time.perf_counter() - t0)
        if _ < 2:
            d.free()
    print(f"upload {words.nbytes/1e9:.2f} GB of packed dna (pageable host memory): {best*1e3:.1f} ms = {words.nbytes/best/1e9:.1f} GB/s")
    t0 = time.perf_counter()
    h = ctx.count_kmers(d, k)
    print(f"first count in a fresh context (the pool's hipMalloc of its work buffers included): {(time.perf_counter()-t0)*1e3:.1f} ms")
    h.free()
    t0 = time.perf_counter()
    h = ctx.count_kmers(d, k)
    t1 = time.perf_counter()
    print(f"count: {(t1-t0)*1e3:.1f} ms; upload + count: {(best + t1 - t0)*1e3:.1f} ms = {(n-k+1)/(best+t1-t0)/1e9:.1f} G k-mers/s PCIe-inclusive")
    m = 100_000_000
    t0 = time.perf_counter()
    gk, gc = h.download(0, m)
    dt = time.perf_counter() - t0
    print(f"download {m} groups in ascending order (gather + 16 B/group to pageable host memory): {dt*1e3:.1f} ms = {m*16/dt/1e9:.1f} GB/s")
    h.free()
    d.free()
