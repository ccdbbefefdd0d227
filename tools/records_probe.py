"""Per-rank critical path of the record exchange on one GPU (no collective): for W ranks, the time of one rank's level 0
over its own rows (dnagpu_sk_records) and of one owner's count over the pieces all W ranks would send it
(dnagpu_count_records), plus the bytes that rank sends / receives.  Second part: the PIPELINED owner of
dnagpu_count_multi_unordered (W ranks on this one device, only owner 0 pulling and counting -- DNAGPU_MULTI_OPT_PROBE_OWNER
-- with its inbound pieces held to an emulated link rate on the transfer stream): critical path = the slowest rank's
level 0 + that owner's phase, for 1 / 2 / 3 bucket groups.
Usage: python tools/records_probe.py [n_bases] [k] [worlds, comma separated] [emulated GB/s per owner, default 350]"""
import importlib
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import load_package  # noqa: E402

pkg = load_package()
sh = importlib.import_module(pkg.__name__ + ".shard_math")
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 3_000_000_000
k = int(sys.argv[2]) if len(sys.argv) > 2 else 31
worlds = [int(x) for x in sys.argv[3].split(",")] if len(sys.argv) > 3 else [1, 2, 4, 8]
link_gbs = float(sys.argv[4]) if len(sys.argv) > 4 else 350.0
seed = 0xD2A0003
rows = n - k + 1
level0 = {}
with pkg.Context(0) as ctx:
    nb = ctx.sk_buckets(rows, k)
    for W in worlds:
        shards = sh.shard_ranges(n, k, W)
        recs, t_l0 = [], []
        for first, cnt, lo, hi in shards:
            d = ctx.synth(seed + lo // 32, hi - lo)
            best = 1e9
            r = None
            for it in range(3):
                if r is not None:
                    r.free()
                ctx.synchronize()
                t0 = time.perf_counter()
                r = ctx.sk_records(d, k, 0, cnt, rows)
                ctx.synchronize()
                best = min(best, time.perf_counter() - t0)
            t_l0.append(best)
            recs.append(r)
            d.free()
        owners = sh.bucket_owner_ranges_weighted([sum(int(r.offsets[b + 1] - r.offsets[b]) for r in recs) for b in range(nb)], W)
        t_cnt, sent, recv, distinct = [], [], [], 0
        for o, (lo_b, hi_b) in enumerate(owners):
            pieces = [(r.device_ptr + 16 * int(r.offsets[b]), int(r.offsets[b + 1] - r.offsets[b]), b)
                      for r in recs for b in range(lo_b, hi_b)]
            recv.append(16 * sum(p[1] for p in pieces))
            best = 1e9
            for it in range(3 if o == 0 else 2):          # (the first call of an owner may allocate: best of two)
                ctx.synchronize()
                t0 = time.perf_counter()
                h = ctx.count_records(pieces, k, rows)
                ctx.synchronize()
                best = min(best, time.perf_counter() - t0)
                if it == 0:
                    distinct += h.distinct
                h.free()
            t_cnt.append(best)
        for ri, r in enumerate(recs):
            lo_b, hi_b = owners[ri]
            sent.append(16 * (r.n_records - int(r.offsets[hi_b] - r.offsets[lo_b])))
            r.free()
        level0[W] = max(t_l0) * 1e3
        print(json.dumps({"world": W, "n_bases": n, "k": k, "buckets": nb, "distinct": distinct,
                          "level0_ms_max": round(max(t_l0) * 1e3, 3), "count_ms_rank0": round(t_cnt[0] * 1e3, 3),
                          "count_ms_max": round(max(t_cnt) * 1e3, 3),
                          "critical_path_ms": round((max(t_l0) + max(t_cnt)) * 1e3, 3),
                          "sent_MB_max": round(max(sent) / 1e6, 1), "recv_MB_max": round(max(recv) / 1e6, 1)}), flush=True)

# ---- the pipelined owner (what dnagpu_count_multi_unordered does), one owner at a time on this device
for W in worlds:
    if W == 1:
        continue
    with pkg.Multi([0] * W, pkg.MULTI_COPY) as m:
        d = m.synth(seed, n)
        m.probe_owner(0)
        for gbs in (0.0, link_gbs):
            m.emulate_link(gbs)
            for parts in (1, 2, 3):
                m.set_parts(parts)
                best = None
                for it in range(3):
                    hs = m.count_unordered(d, k)
                    t = m.last_times()
                    for h in hs:
                        h.free()
                    if best is None or t["count_ms"] < best["count_ms"]:
                        best = t
                print(json.dumps({"world": W, "pipelined_owner": 0, "parts": parts, "emulated_link_GBs": gbs,
                                  "owner_phase_ms": round(best["count_ms"], 3), "exchange_ms": round(best["exchange_ms"], 3),
                                  "hidden_ms": round(best["hidden_ms"], 3), "level0_ms_max": round(level0[W], 3),
                                  "critical_path_ms": round(level0[W] + best["count_ms"], 3),
                                  "MB_in": round(best["bytes_moved"] / 1e6, 1)}), flush=True)
        m.dna_free(d)
