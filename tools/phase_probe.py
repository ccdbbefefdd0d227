"""Per-phase device times of the count at several sizes (HIP events inside the library).
Usage: python tools/phase_probe.py [n_bases ...]   -> one line per phase, JSON at the end."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import load_package  # noqa: E402


def main():
    pkg = load_package()
    sizes = [int(float(a)) for a in sys.argv[1:]] or [100_000_000, 1_000_000_000]
    out = []
    with pkg.Context(0) as ctx:
        ctx.set_profiling(True)
        for n in sizes:
            for k in (31, 21):
                d = ctx.synth(0xD2A0003, n)
                ctx.synchronize()
                best = None
                for it in range(3):
                    t0 = time.perf_counter()
                    h = ctx.count_kmers(d, k)
                    dt = time.perf_counter() - t0
                    phases = ctx.last_phase_times()
                    distinct = h.distinct
                    h.free()
                    if best is None or dt < best[0]:
                        best = (dt, phases, distinct)
                d.free()
                dt, phases, distinct = best
                rec = {"n_bases": n, "k": k, "wall_ms": dt * 1e3, "distinct": distinct,
                       "gkmers_per_s": (n - k + 1) / dt / 1e9, "phases": phases,
                       "device_bytes": ctx.device_bytes()}
                out.append(rec)
                print(f"n={n:.3e} k={k}: {dt*1e3:.2f} ms wall, {rec['gkmers_per_s']:.2f} G k-mers/s, "
                      f"distinct={distinct}, pool={ctx.device_bytes()/2**30:.1f} GiB", flush=True)
                for name, ms in phases:
                    print(f"    {name:18s} {ms:9.3f} ms", flush=True)
            ctx.trim()
    print(json.dumps(out))


if __name__ == "__main__":
    main()
