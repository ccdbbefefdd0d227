import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import oracle as orc
from __graft_entry__ import load_package
pkg = load_package()
cases = [(100_000, 1000), (3_000_000, 1000), (2_000_000, 7), (500_000, 250_000)]
with pkg.Context(0) as ctx:
    for n, motif in cases:
        d = ctx.synth(4, n, motif_len=motif)
        words = d.download()
        for k in (5, 12, 21, 31, 32):
            print("case", n, motif, k, flush=True)
            h = ctx.count_kmers(d, k)
            print("   distinct", h.distinct, [(a, round(b, 3)) for a, b in ctx.last_phase_times()], flush=True)
            ok, oc = orc.count_kmers(words, n, k)
            gk, gc = h.download()
            print("   match", np.array_equal(gk, ok) and np.array_equal(gc, oc), len(ok), flush=True)
            h.free()
        d.free()
