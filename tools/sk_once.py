"""One unordered (super-k-mer) count for profilers: python tools/sk_once.py [n_bases] [k] [iters]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import load_package  # noqa: E402

pkg = load_package()
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1_000_000_000
k = int(sys.argv[2]) if len(sys.argv) > 2 else 31
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 2
with pkg.Context(0) as ctx:
    d = ctx.synth(0xD2A0003, n, motif_len=int(sys.argv[4]) if len(sys.argv) > 4 else 0)
    for _ in range(iters):
        h = ctx.count_kmers_unordered(d, k)
        print("distinct", h.distinct)
        h.free()
