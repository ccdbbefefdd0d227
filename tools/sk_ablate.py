"""Timing ablations of the super-k-mer engine (diagnostic build, DNAGPU_DEBUG_SK bits; results invalid).
Usage: DNAGPU_DEBUG_SK=<bits> python tools/sk_ablate.py [n_bases] [k]"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import load_package  # noqa: E402

pkg = load_package()
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1_000_000_000
k = int(sys.argv[2]) if len(sys.argv) > 2 else 31
with pkg.Context(0) as ctx:
    d = ctx.synth(0xD2A0003, n)
    ctx.set_profiling(True)
    for it in range(3):
        try:
            h = ctx.count_kmers_unordered(d, k)
            h.free()
            err = None
        except Exception as e:
            err = str(e)[:80]
    print(json.dumps({"dbg": os.environ.get("DNAGPU_DEBUG_SK", "0"), "err": err,
                      "phases_ms": {a: round(b, 3) for a, b in ctx.last_phase_times()}}), flush=True)
