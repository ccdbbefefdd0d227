#!/usr/bin/env python3
"""bench.py -- k-mers/sec for the k=31 count over 3 Gbase synthetic (BASELINE.json metric).

  python bench.py --gpus N --steps K --warmup W            (default: --config 4, the headline)
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...
  python bench.py --config {2,3,4,5} [--motif M] [--pattern P]   the other BASELINE.json configs / repeat-rich variants

A step = one full GROUP BY count over the whole synthetic sequence (extraction fused in), input
already resident in HBM.  N=1: the single-GPU path (dnagpu_count_kmers_unordered) on all 3 Gbase.  N>1: the same
3 Gbase sharded by contiguous chunk over the ranks (strong scaling), three ways to start it:

  python bench.py --gpus N                       ONE process drives N GPUs through the C-ABI (dnagpu_multi_init +
                                                 dnagpu_count_multi_unordered: every rank cuts the super-k-mer records of
                                                 its own rows, owners pull their buckets' pieces over xGMI and count them
                                                 while later pieces are in flight).  No torch.  This is the call a
                                                 PostgreSQL backend makes (one backend process, nothing PARALLEL SAFE:
                                                 dna--1.0.sql:188-195).  On a box with fewer than N devices the ranks
                                                 share devices and the line says "rehearsal": true.
  python bench.py --gpus N --launcher torchrun   spawns torch.distributed.run (one process per GPU, RCCL) as a CHILD
                                                 before this process imports torch or touches a GPU, relays its line/rc
  python -m torch.distributed.run ... bench.py --gpus N     (WORLD_SIZE in the environment) one process per GPU: the
                                                 records travel in one RCCL all-to-all (sharded.py)
Rank 0 prints ONE JSON line.

Besides the contract fields the line carries
  roofline      the dominant kernel of the step: algorithmic bytes / its device time (HIP events on
                the library's stream) against the 8 TB/s HBM peak
  cpu_baseline  the CPU oracle (faithful restatement of the reference's per-base loops + hash
                aggregate) timed on this box's host cores on a bounded sample of the same stream
  job_roofline  SURVEY.md 8(d)'s two whole-job fractions (read_fraction, alg_fraction)
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)
CPU_SAMPLE_BASES = 24_000_000  # bounded sample for the CPU baseline (about 10-20 s on one core)

# BASELINE.json configs: synthetic packed words, word w = splitmix64(seed + w)
CONFIGS = {
    2: {"n_bases": 100_000_000, "k": 21, "seed": 0xD2A0001, "kind": "count"},
    3: {"n_bases": 248_956_422, "k": 31, "seed": 0xD2A0002, "kind": "count"},
    4: {"n_bases": 3_000_000_000, "k": 31, "seed": 0xD2A0003, "kind": "count"},      # the headline (default)
    5: {"n_bases": 100_000_000, "k": 21, "seed": 0xD2A0001, "kind": "filter", "pattern": "NNNNNNNNNNWSNNNNNNNNN"},
    # not a BASELINE.json config: the reference's second counting shape (test.sql:140-150, GROUP BY over a TABLE of
    # sequences) at read scale -- 10^7 reads of 150 bases in one packed stream (dnagpu_count_kmers_batch)
    6: {"n_bases": 1_500_000_000, "k": 31, "seed": 0xD2A0006, "kind": "count", "read_len": 150},
}
HEADLINE_METRIC = "k-mers/sec for k=31 count over 3 Gbase synthetic; % of HBM-read roofline"

# algorithmic HBM bytes of each phase per k-mer (n) / per distinct k-mer (d); DESIGN.md "kernels"
# (n = this rank's k-mers, d = its distinct k-mers, N = k-mers of the whole sequence: a sharded rank sweeps
# all of the packed sequence at level 0 and keeps its own key range)
PHASE_BYTES = {
    "hist0": lambda n, d, N: 0.25 * N,              # packed input only
    "scatter0": lambda n, d, N: 0.25 * N + 8.0 * n,  # packed input in, keys out
    "hist": lambda n, d, N: 8.0 * n,                # keys in
    "scatter": lambda n, d, N: 16.0 * n,            # keys in, keys out
    "leaves": lambda n, d, N: 8.0 * n + 12.0 * d,   # keys in, (u64 key, u32 count) groups out
    "dense": lambda n, d, N: 0.25 * N,              # short k-mers: the packed input per pass
    # super-k-mer engine (dnagpu_count_kmers_unordered, k >= 21): 16-byte records of ~9 k-mers at k = 31 (runs sharing a
    # minimizer of m = 15 bases, 13 for k = 21 / 22; mean run length (w + 1) / 2 at window w = k - m + 1) instead of 8-byte keys
    "sk_hist0": lambda n, d, N: 0.25 * N,
    "sk_scatter0": lambda n, d, N: 0.25 * N + 16.0 * n / SK_RUN,
    "sk_hist1": lambda n, d, N: 16.0 * n / SK_RUN,
    "sk_scatter1": lambda n, d, N: 32.0 * n / SK_RUN,
    "sk_regroup": lambda n, d, N: 32.0 * n / SK_RUN,
    "sk_count": lambda n, d, N: 16.0 * n / SK_RUN + 12.0 * d,   # records in, (u64 key, u32 count) groups out
}
SK_RUN = 9.0   # k-mers per record at k = 31 (window of 17 m-mers); set per run from k in main()


def phase_kind(name):
    if name == "leaves":
        return "leaves"
    if name in PHASE_BYTES and name.startswith("sk_"):
        return name
    if name.endswith("_hist"):
        return "hist0" if name.startswith("level0") else "hist"
    if name.endswith("_scatter"):
        return "scatter0" if name.startswith("level0") else "scatter"
    return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--config", type=int, default=4, choices=sorted(CONFIGS),
                    help="BASELINE.json config: 2 (k=21 count, 100 Mbase), 3 (k=31 count, chr1 scale), "
                         "4 (k=31 count, 3 Gbase: the headline, default), 5 (qkmer @> fused into k=21 extraction); "
                         "6 (not in BASELINE.json: k=31 count over a table of 10^7 reads of 150 bases)")
    ap.add_argument("--read-len", type=int, default=None, help="config 6: bases per read")
    ap.add_argument("--n-bases", type=float, default=None, help="override the config's sequence length")
    ap.add_argument("--k", type=int, default=None, help="override the config's k")
    ap.add_argument("--motif", type=int, default=0,
                    help="repeat-rich variant (SURVEY.md 8(d)): tile the first MOTIF bases over the second half")
    ap.add_argument("--keys-only", action="store_true",
                    help="config 5: write the matching kmers only (what the SQL statement returns), not their positions too")
    ap.add_argument("--table-host-starts", action="store_true",
                    help="config 6: hand the sequence starts over as a host array with every count (dnagpu_count_kmers_batch: "
                         "8 bytes per sequence cross the bus inside the timed step) instead of making them resident once")
    ap.add_argument("--genome-like", action="store_true",
                    help="count configs, single GPU: instead of uniform random bases a genome-LIKE sequence generated on the host "
                         "(tools/genome_like.py: an Alu-like family, exact segmental duplications, microsatellites) and uploaded")
    ap.add_argument("--pattern", type=str, default=None, help="config 5: the qkmer pattern (length k)")
    ap.add_argument("--engine", choices=["auto", "tree"], default="auto",
                    help="count configs: auto = dnagpu_count_kmers_unordered (GROUP BY semantics: group order "
                         "unspecified; super-k-mer partitioning for k >= 21 on long sequences), tree = "
                         "dnagpu_count_kmers (groups in ascending key order, MSD radix tree)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--launcher", choices=["one-process", "torchrun"], default="one-process",
                    help="--gpus N > 1 without WORLD_SIZE in the environment: one-process = this process drives all N GPUs "
                         "through the C-ABI (default); torchrun = spawn torch.distributed.run as a child, one process per GPU")
    ap.add_argument("--exchange", choices=["copy", "rccl"], default="copy",
                    help="one-process path, record exchange: copy = owners pull with peer copies (default), rccl = "
                         "ncclSend / ncclRecv per piece (needs distinct devices and librccl)")
    ap.add_argument("--parts", type=int, default=0,
                    help="one-process path: bucket groups per owner of the pipelined record exchange (0 = library default)")
    ap.add_argument("--emulate-link-gbs", type=float, default=0.0,
                    help="one-process REHEARSAL on shared devices: hold inbound pieces to this many GB/s per owner")
    args = ap.parse_args()
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        if args.launcher == "torchrun":
            raise SystemExit(spawn_torchrun(args.gpus))
        return main_one_process(args)

    cfg = dict(CONFIGS[args.config])
    if args.n_bases is not None:
        cfg["n_bases"] = int(args.n_bases)
    if args.k is not None:
        cfg["k"] = args.k
    if args.pattern is not None:
        cfg["pattern"] = args.pattern
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    n_bases, k, seed = cfg["n_bases"], cfg["k"], cfg["seed"]
    n_kmers = n_bases - k + 1
    global SK_RUN
    SK_RUN = (max(k - (15 if k >= 23 else 13 if k >= 21 else 12) + 1, 1) + 1) / 2.0
    is_filter = cfg["kind"] == "filter"
    if is_filter and world > 1:
        raise SystemExit("config 5 is a single-GPU workload (BASELINE.json)")

    import torch
    from __graft_entry__ import load_package
    pkg = load_package()

    dist = None
    if world > 1:
        import torch.distributed as dist
        # nccl (= RCCL over xGMI) is the product path.  BENCH_BACKEND=gloo with BENCH_ONE_GPU=1 is a
        # rehearsal rig only: all ranks share GPU 0 and collectives are staged through the host.
        backend = os.environ.get("BENCH_BACKEND", "nccl")
        if os.environ.get("BENCH_ONE_GPU") == "1":
            local_rank = 0
        torch.cuda.set_device(local_rank)
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"

    ctx = pkg.Context(local_rank)
    ctx.set_profiling(True)

    def sync_all():
        if dist is not None:
            dist.barrier()
        ctx.synchronize()
        torch.cuda.synchronize()

    phases_acc = {}
    distinct = [0]
    matches = [0]
    extra = {}
    sorted_result = [True]

    if is_filter:
        import ctypes as C
        dna = ctx.synth(seed, n_bases, motif_len=args.motif)
        flt = pkg.Filter.contains(cfg["pattern"])
        # outputs (keys and positions) stay in device memory, sized for every row
        kb = ctx.buffer_alloc(n_kmers * 8)
        pb = None if args.keys_only else ctx.buffer_alloc(n_kmers * 8)

        def step():
            matches[0] = ctx.count_matches_device(dna, k, flt, 0, n_kmers, C.c_void_p(kb), C.c_void_p(pb), n_kmers)
            for name, ms in ctx.last_phase_times():
                phases_acc.setdefault(name, []).append(ms)
    elif world == 1 and cfg.get("read_len"):
        import numpy as np
        read_len = args.read_len or cfg["read_len"]
        n_reads = n_bases // read_len
        n_bases = n_reads * read_len
        dna = ctx.synth(seed, n_bases, motif_len=args.motif)
        starts = np.arange(n_reads + 1, dtype=np.uint64) * np.uint64(read_len)
        n_kmers = n_reads * max(read_len - k + 1, 0)                 # the rows of the table: every read's own
        extra["table"] = {"reads": n_reads, "read_len": read_len,
                          "boundaries": "host array per call (dnagpu_count_kmers_batch)" if args.table_host_starts else
                                        "resident (dnagpu_dna_set_sequences once, dnagpu_count_kmers_table per step)"}
        if not args.table_host_starts:
            dna.set_sequences(starts)              # inputs resident in HBM when the timed region starts

        def count_fn(dna_, k_):
            return ctx.count_kmers_batch(dna_, starts, k_) if args.table_host_starts else ctx.count_kmers_table(dna_, k_)

        def step():
            h = count_fn(dna, k)
            sorted_result[0] = h.is_sorted
            distinct[0] = h.distinct
            for name, ms in ctx.last_phase_times():
                phases_acc.setdefault(name, []).append(ms)
            h.free()
    elif world == 1:
        if args.genome_like:
            sys.path.insert(0, os.path.join(ROOT, "tools"))
            import genome_like
            dna = ctx.upload(genome_like.packed_words(n_bases, seed), n_bases)
            extra["input"] = "genome-like (tools/genome_like.py: Alu-like copies 10 %, segmental duplications 2 %, microsatellites 1 %)"
        else:
            dna = ctx.synth(seed, n_bases, motif_len=args.motif)

        count_fn = ctx.count_kmers if args.engine == "tree" else ctx.count_kmers_unordered

        def step():
            h = count_fn(dna, k)
            sorted_result[0] = h.is_sorted
            distinct[0] = h.distinct
            for name, ms in ctx.last_phase_times():
                phases_acc.setdefault(name, []).append(ms)
            h.free()
    else:
        import importlib
        if args.motif:
            raise SystemExit("--motif is a single-GPU option")
        sh = importlib.import_module(pkg.__name__ + ".sharded")
        engine = sh.GpuEngine(pkg, ctx, torch.device("cuda", local_rank))
        state = {"chunk": None}
        # long k-mers, any group order (the single-GPU default's rule): every rank cuts the super-k-mer records of its own
        # rows, the coarse buckets travel to their owners (one all-to-all of 1.8 B per k-mer), the owners count them
        use_records = args.engine != "tree" and k >= 21
        extra["exchange"] = ("records: own rows -> super-k-mer records -> all-to-all by coarse bucket -> count" if use_records else
                             "sequence: all-gather of the packed chunks -> every rank counts the key range it owns")

        def step():
            if use_records:
                h, state["chunk"] = sh.count_sharded_exchange_records(engine, seed, n_bases, k, rank, world, state["chunk"],
                                                                      parts=max(args.parts, 1))
            else:
                # resident input = this rank's word chunk of the packed sequence; the step all-gathers
                # the chunks (RCCL) and counts the keys this rank owns over the whole sequence
                h, state["chunk"] = sh.count_sharded(engine, seed, n_bases, k, rank, world, state["chunk"])
            distinct[0] = h.distinct
            per_step = {}                          # (a phase name appears once per bucket group of the exchange: their sum)
            for name, ms in (engine.phase_times() if use_records else ctx.last_phase_times()):
                per_step[name] = per_step.get(name, 0.0) + ms
            for name, ms in per_step.items():
                phases_acc.setdefault(name, []).append(ms)
            h.free()

    for _ in range(args.warmup):
        step()
    phases_acc.clear()
    sync_all()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    sync_all()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        red_dev = "cuda" if dist.get_backend() == "nccl" else "cpu"
        t = torch.tensor([elapsed], dtype=torch.float64, device=red_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        dsum = torch.tensor([distinct[0]], dtype=torch.int64, device=red_dev)
        dist.all_reduce(dsum)
        distinct[0] = int(dsum.item())

    # ---- outside the timed region: one more count whose histogram is digested on the device and compared with the CPU
    # oracle's digest of the same workload (every rank's share summed: the histograms are disjoint)
    digest = None
    if not is_filter:
        if world == 1:
            hv = count_fn(dna, k)
        elif use_records:
            hv, state["chunk"] = sh.count_sharded_exchange_records(engine, seed, n_bases, k, rank, world, state["chunk"],
                                                                   parts=max(args.parts, 1))
        else:
            hv, state["chunk"] = sh.count_sharded(engine, seed, n_bases, k, rank, world, state["chunk"])
        mine = [int(x) & 0xFFFFFFFFFFFFFFFF for x in hv.summary()]
        hv.free()
        if dist is not None:
            # (int64 tensors: the checksum's wrapping sum is reassembled from two 32-bit halves)
            parts_ = torch.tensor([[v >> 32, v & 0xFFFFFFFF] for v in mine], dtype=torch.int64, device=red_dev)
            dist.all_reduce(parts_)
            mine = [((int(hi) << 32) + int(lo)) & 0xFFFFFFFFFFFFFFFF for hi, lo in parts_.tolist()]
        digest = digest_check(args.config if not args.genome_like else 0, n_bases, k, seed, args.motif, mine, rows=n_kmers)

    if rank == 0 and world == 1 and not is_filter and not sorted_result[0]:
        extra["sorted_view_ms"] = None
        extra["group_order"] = ("unspecified (bucket order of the super-k-mer engine; PostgreSQL's GROUP BY order is "
                                "unspecified too, test.sql:95-104); --engine tree gives ascending keys")
    elif rank == 0 and world == 1 and not is_filter:
        # outside the timed region: what the ascending-key view of the same histogram costs on the device
        # (dnagpu_hist_sorted_view: segment directory -> two dense uint64 arrays); groups are stored in
        # completion order, PostgreSQL's GROUP BY order is unspecified too (test.sql:95-104)
        try:
            import ctypes as C
            h = ctx.count_kmers(dna, k)
            d = h.distinct
            vk, vc = ctx.buffer_alloc(max(d, 1) * 8), ctx.buffer_alloc(max(d, 1) * 8)
            h.sorted_view_device(C.c_void_p(vk), C.c_void_p(vc))        # builds the directory prefix once
            ctx.synchronize()
            t1 = time.perf_counter()
            h.sorted_view_device(C.c_void_p(vk), C.c_void_p(vc))
            extra["sorted_view_ms"] = round((time.perf_counter() - t1) * 1e3, 3)
            ctx.buffer_free(vk)
            ctx.buffer_free(vc)
            h.free()
        except Exception as e:                                          # never lose the line over the extra
            extra["sorted_view_ms"] = None
            extra["sorted_view_error"] = repr(e)[:200]

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        value = n_kmers * args.steps / elapsed
        means = {name: sum(v) / len(v) for name, v in phases_acc.items()}
        b_in = 8 * ((n_bases + 31) // 32)
        t_step = elapsed / args.steps
        default_workload = (args.config == 4 and n_bases == CONFIGS[4]["n_bases"] and k == CONFIGS[4]["k"]
                            and not args.motif)
        roofline = None
        if is_filter:
            # dominant kernel: the write sweep (tests every row again, cuts and stores the matching keys + positions)
            dom = "filter_write"
            out_bytes = (8 if args.keys_only else 16) * matches[0]
            if dom in means and means[dom] > 0:
                alg_bytes = b_in + out_bytes
                achieved = alg_bytes / (means[dom] * 1e-3) / 1e9
                roofline = {"bound": "hbm", "kernel": dom, "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS,
                            "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4),
                            "alg_bytes_per_launch": int(alg_bytes), "kernel_ms": round(means[dom], 4),
                            "traffic": load_traffic(f"config5:{dom}")}
            job = {"read_fraction": round(b_in / t_step / 1e9 / HBM_PEAK_GBS, 5),
                   "alg_fraction": round((b_in + out_bytes) / t_step / 1e9 / HBM_PEAK_GBS, 4),
                   "alg_bytes_per_row": round((b_in + out_bytes) / n_kmers, 3),
                   "note": ("output = the matching kmers, 8 B per match (what the SQL statement returns: rows of kmer)"
                            if args.keys_only else
                            "compulsory output counted as 16 B per match (key + position, both written)")}
            metric = (f"rows/sec scanned, qkmer @> fused into the k={k} extraction over {n_bases} synthetic bases; "
                      "% of HBM-read roofline")
            workload = (f"config 5: generate_kmers(dna,{k}) WHERE '{cfg['pattern']}' @> kmer over {n_bases} synthetic bases "
                        f"(splitmix64 seed {seed:#x}{', motif ' + str(args.motif) if args.motif else ''}), "
                        f"{'keys' if args.keys_only else 'keys + positions'} to device memory, single GPU")
            unit = "rows/s"
        else:
            kern = {n_: m for n_, m in means.items() if phase_kind(n_)}
            dom = max(kern, key=kern.get) if kern else None
            if dom:
                per_rank_n = n_kmers / world
                per_rank_d = distinct[0] / world
                alg_bytes = PHASE_BYTES[phase_kind(dom)](per_rank_n, per_rank_d, n_kmers)
                achieved = alg_bytes / (means[dom] * 1e-3) / 1e9
                roofline = {"bound": "hbm", "kernel": dom, "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS,
                            "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4),
                            "alg_bytes_per_launch": int(alg_bytes), "kernel_ms": round(means[dom], 3),
                            # PMC traffic is collected for the default single-GPU workload only
                            "traffic": load_traffic(dom) if (world == 1 and default_workload) else None}
            job = job_fractions(b_in, distinct[0], n_kmers, t_step, world)
            metric = HEADLINE_METRIC if (args.config == 4 and n_bases == CONFIGS[4]['n_bases'] and k == CONFIGS[4]['k']) else \
                f"k-mers/sec for k={k} count over {n_bases} synthetic bases; % of HBM-read roofline"
            eng = "" if world > 1 else (", ordered groups (MSD radix tree)" if sorted_result[0] else ", unordered groups (super-k-mer partitioning)")
            tbl = (f" as a table of {extra['table']['reads']} sequences of {extra['table']['read_len']} bases "
                   "(dnagpu_count_kmers_table: no k-mer spans two sequences)") if "table" in extra else ""
            workload = (f"config {args.config}: k={k} count over {n_bases} synthetic bases{tbl} (splitmix64 seed {seed:#x}"
                        f"{', motif ' + str(args.motif) if args.motif else ''}){eng}, "
                        f"{'single GPU' if world == 1 else f'sharded over {world} GPUs, ' + extra.get('exchange', '')}")
            unit = "k-mers/s"
        line = {
            "metric": metric,
            "value": value, "unit": unit, "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "u64", "data": "synthetic",
            "config": {"workload": workload, "n_bases": n_bases, "k": k,
                       **({"pattern": cfg["pattern"], "matches": matches[0]} if is_filter else {"distinct": distinct[0]}),
                       **({"motif": args.motif} if args.motif else {})},
            "roofline": roofline,
            "job_roofline": job,
            "phases_ms": {n_: round(m, 4) for n_, m in means.items()},
        }
        line.update(extra)
        if digest is not None:
            line.update(digest)
        if world > 1:
            line["launcher"] = "torchrun"
            line["exchange_transport"] = (f"{dist.get_backend()} " + ("all_to_all_single" if use_records and max(args.parts, 1) == 1 else
                                          ("batch_isend_irecv" if use_records else "all_gather")))
            line["rccl_ranks"] = world if dist.get_backend() == "nccl" else 0
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline_filter(cfg, seed) if is_filter else cpu_baseline(k, seed)
        print(json.dumps(line), flush=True)

    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    ctx.close()
    if digest is not None and digest_failed(digest):
        raise SystemExit(3)                       # the groups differ from the CPU oracle's: the number above is void


def spawn_torchrun(n_gpus):
    """--launcher torchrun: the process-per-GPU path as a CHILD (this process has not imported torch nor touched a GPU,
    and never execs: it relays the child's output and exit code)."""
    import socket
    import subprocess
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    argv = [a for a in sys.argv[1:]]
    # drop "--launcher torchrun" from the child's arguments (WORLD_SIZE in its environment selects the path there)
    out = []
    skip = False
    for a in argv:
        if skip:
            skip = False
            continue
        if a == "--launcher":
            skip = True
            continue
        if a.startswith("--launcher="):
            continue
        out.append(a)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n_gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + out
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.call(cmd, env=env)


def main_one_process(args):
    """python bench.py --gpus N: N ranks driven from THIS process through the C-ABI (dnagpu_multi_*), no torch.
    The same JSON contract; `phases_ms` = rank 0's device phases, `exchange` = the host clock of the call's phases."""
    cfg = dict(CONFIGS[args.config])
    if args.n_bases is not None:
        cfg["n_bases"] = int(args.n_bases)
    if args.k is not None:
        cfg["k"] = args.k
    if cfg["kind"] == "filter":
        raise SystemExit("config 5 is a single-GPU workload (BASELINE.json)")
    if args.motif:
        raise SystemExit("--motif is a single-GPU option")
    n_bases, k, seed = cfg["n_bases"], cfg["k"], cfg["seed"]
    n_kmers = n_bases - k + 1
    W = args.gpus
    global SK_RUN
    SK_RUN = (max(k - (15 if k >= 23 else 13 if k >= 21 else 12) + 1, 1) + 1) / 2.0

    from __graft_entry__ import load_package
    pkg = load_package()
    n_dev = pkg.device_count()
    if n_dev < 1:
        raise SystemExit("no HIP device")
    rehearsal = n_dev < W
    devices = [r % n_dev for r in range(W)]
    # RCCL (all-gather / reduce of the ordered paths) when the devices are distinct and the library loads, else peer
    # copies; the record exchange itself always pulls with peer copies
    multi = pkg.Multi(devices, pkg.MULTI_AUTO)
    if args.parts:
        multi.set_parts(args.parts)
    if args.emulate_link_gbs:
        multi.emulate_link(args.emulate_link_gbs)
    for c in multi.ranks:
        c.set_profiling(True)
    mdna = multi.synth(seed, n_bases)
    use_records = args.engine != "tree" and k >= 21
    count_fn = multi.count_unordered if use_records else multi.count
    if args.exchange == "rccl":
        multi.set_exchange_rccl(1)                 # (fails loudly when no communicator exists: shared devices, no librccl)

    def sync_all():
        for c in multi.ranks:
            c.synchronize()

    phases_acc, xacc = {}, {}
    distinct = [0]

    def step():
        hs = count_fn(mdna, k)
        distinct[0] = sum(h.distinct for h in hs)
        per_step = {}                              # (the pipelined exchange counts an owner's buckets group by group: a phase
        for name, ms in (multi.rank_phase_times(0) if use_records else multi.ranks[0].last_phase_times()):
            per_step[name] = per_step.get(name, 0.0) + ms      # (... name then appears once per group: their SUM is the step's)
        for name, ms in per_step.items():
            phases_acc.setdefault(name, []).append(ms)
        if use_records:
            for name, v in multi.last_times().items():
                xacc.setdefault(name, []).append(v)
        for h in hs:
            h.free()

    for _ in range(args.warmup):
        step()
    phases_acc.clear()
    xacc.clear()
    sync_all()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    sync_all()
    elapsed = time.perf_counter() - t0

    # outside the timed region: one more count, every rank's histogram digested on its device, the sums against the CPU
    # oracle's digest of the same workload
    hs = count_fn(mdna, k)
    sums = [0, 0, 0, 0]
    for h in hs:
        for i, v in enumerate(h.summary()):
            sums[i] = (sums[i] + int(v)) & 0xFFFFFFFFFFFFFFFF
        h.free()
    digest = digest_check(args.config, n_bases, k, seed, 0, sums)
    exchange_transport = multi.exchange_transport

    ms_per_step = elapsed / args.steps * 1e3
    t_step = elapsed / args.steps
    value = n_kmers * args.steps / elapsed
    means = {name: sum(v) / len(v) for name, v in phases_acc.items()}
    b_in = 8 * ((n_bases + 31) // 32)
    kern = {n_: m for n_, m in means.items() if phase_kind(n_)}
    dom = max(kern, key=kern.get) if kern else None
    roofline = None
    if dom:
        alg_bytes = PHASE_BYTES[phase_kind(dom)](n_kmers / W, distinct[0] / W, n_kmers)
        achieved = alg_bytes / (means[dom] * 1e-3) / 1e9
        roofline = {"bound": "hbm", "kernel": dom, "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(achieved / HBM_PEAK_GBS, 4), "alg_bytes_per_launch": int(alg_bytes),
                    "kernel_ms": round(means[dom], 3), "traffic": None,
                    "note": "rank 0's share of the work (1/N of the rows); PMC traffic is collected for the single-GPU workload only"}
    xm = {name: sum(v) / len(v) for name, v in xacc.items()}
    exchange = None
    if use_records:
        exchange = {"what": "records: own rows -> super-k-mer records -> every coarse bucket's pieces to its owner ("
                            + ("one ncclSend / ncclRecv per piece, a group call per bucket group" if exchange_transport == "rccl-sendrecv"
                               else "the owner pulls them: peer copies of 16-byte records")
                            + ") -> counted group by group while later groups are in flight",
                    "parts": int(xm.get("parts", 0)), "records_ms": round(xm.get("records_ms", 0), 3),
                    "exchange_ms": round(xm.get("exchange_ms", 0), 3), "hidden_ms": round(xm.get("hidden_ms", 0), 3),
                    "owners_ms": round(xm.get("count_ms", 0), 3), "call_ms": round(xm.get("total_ms", 0), 3),
                    "MB_moved_per_step": round(xm.get("bytes_moved", 0) / 1e6, 1),
                    **({"emulated_link_GBs_per_owner": args.emulate_link_gbs} if args.emulate_link_gbs else {})}
    else:
        exchange = {"what": "sequence: all-gather of the packed chunks -> every rank counts the key range it owns"}
    line = {
        "metric": HEADLINE_METRIC if (args.config == 4 and n_bases == CONFIGS[4]["n_bases"] and k == CONFIGS[4]["k"]) else
        f"k-mers/sec for k={k} count over {n_bases} synthetic bases; % of HBM-read roofline",
        # (a rehearsal -- ranks sharing devices -- reports the devices it really used, so that the line cannot be read as
        # a scaling point; "ranks" is always the N that was asked for)
        "value": value, "unit": "k-mers/s", "n_gpus": len(set(devices)), "ranks": W, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
        "dtype": "u64", "data": "synthetic",
        "config": {"workload": f"config {args.config}: k={k} count over {n_bases} synthetic bases (splitmix64 seed {seed:#x}), "
                               f"sharded by contiguous chunk over {W} ranks driven by one process through the C-ABI "
                               f"(dnagpu_count_multi{'_unordered' if use_records else ''})",
                   "n_bases": n_bases, "k": k, "distinct": distinct[0]},
        "launcher": "one-process", "devices": devices, "distinct_devices": len(set(devices)),
        # what the timed path moved its data with -- NOT what is merely available (rccl_available)
        "exchange_transport": exchange_transport,
        "rccl_ranks": multi.rccl_ranks if exchange_transport.startswith("rccl") else 0,
        "rccl_available": multi.transport == "rccl", "rehearsal": rehearsal,
        "roofline": roofline,
        "job_roofline": job_fractions(b_in, distinct[0], n_kmers, t_step, W),
        "phases_ms": {n_: round(m, 4) for n_, m in means.items()},
        "exchange": exchange,
    }
    if rehearsal:
        line["rehearsal_note"] = (f"{W} ranks share {n_dev} device(s): the code path of the N-GPU run (chunk residency, halo "
                                  "word, per-rank records, owners' pulls, pipelined count), NOT a scaling measurement")
    line.update(digest)
    print(json.dumps(line), flush=True)
    multi.dna_free(mdna)
    multi.close()
    if digest_failed(digest):
        raise SystemExit(3)                       # the groups differ from the CPU oracle's: the number above is void


def digest_key(config, n_bases, k, motif):
    """key of tests/golden/config_digests.json for this workload, or None when the oracle never counted it"""
    base = CONFIGS.get(config)
    if not base or base["kind"] != "count" or n_bases != base["n_bases"] or k != base["k"]:
        return None
    return f"{config}m{motif}" if motif else str(config)


def digest_check(config, n_bases, k, seed, motif, got, rows=None):
    """got = (total, distinct, unique, checksum) summed over the ranks' histograms (dnagpu_hist_summary: the checksum is a
    wrapping sum over groups, so disjoint histograms add up).  Compared with the CPU oracle's digest of the same workload
    (tools/make_digests.py -> tests/golden/config_digests.json).  -> the fields for the JSON line; digest_ok is None when
    the oracle holds no digest for this workload (then only total == rows is checked: digest_total_ok)."""
    total, distinct, unique, checksum = (int(x) for x in got)
    out = {"digest": {"total": total, "distinct": distinct, "unique": unique, "checksum": checksum}}
    out["digest_total_ok"] = total == (n_bases - k + 1 if rows is None else rows)
    key = digest_key(config, n_bases, k, motif)
    want = None
    if key is not None:
        try:
            # (BENCH_DIGESTS: the test of this check points it at a doctored copy to see it fail)
            with open(os.environ.get("BENCH_DIGESTS") or os.path.join(ROOT, "tests", "golden", "config_digests.json")) as f:
                want = json.load(f).get(key)
        except Exception:
            want = None
    if want is None or int(want["seed"]) != seed:
        out["digest_ok"] = None
        out["digest_source"] = "no oracle digest for this workload: only sum(count) == rows was checked"
        return out
    out["digest_ok"] = all(int(want[f]) == v for f, v in
                           (("total", total), ("distinct", distinct), ("unique", unique), ("checksum", checksum)))
    out["digest_source"] = f"tests/golden/config_digests.json[{key!r}] (CPU oracle, tools/make_digests.py)"
    if not out["digest_ok"]:
        out["digest_expected"] = {f: int(want[f]) for f in ("total", "distinct", "unique", "checksum")}
    return out


def digest_failed(d):
    return d.get("digest_ok") is False or d.get("digest_total_ok") is False


def job_fractions(b_in, distinct, n_kmers, t_step, world):
    """SURVEY.md 8(d)'s whole-job fractions.  alg_fraction prices a group at SURVEY's 16 B (8-byte key + 8-byte count);
    the kernels write 12 B (uint32 counts, widened on download): both are reported."""
    per_gpu = t_step * 1e9 * HBM_PEAK_GBS * world
    return {"read_fraction": round(b_in / per_gpu, 5),
            "alg_fraction_16B": round((b_in + 16 * distinct) / per_gpu, 4),
            "alg_fraction_12B": round((b_in + 12 * distinct) / per_gpu, 4),
            "alg_fraction": round((b_in + 16 * distinct) / per_gpu, 4),
            "alg_bytes_per_kmer": round((b_in + 16 * distinct) / n_kmers, 3),
            "alg_bytes_per_kmer_written": round((b_in + 12 * distinct) / n_kmers, 3)}


def load_traffic(kernel):
    """HBM bytes per launch of the dominant kernel from the committed PMC profile, if one matches."""
    p = os.path.join(ROOT, "profiles", "traffic_latest.json")
    try:
        with open(p) as f:
            t = json.load(f)
        return t.get(kernel)
    except Exception:
        return None


def cpu_baseline_filter(cfg, seed):
    """Config 5's CPU leg: the oracle's per-row contains() loop (nucleotide_matches per base, dna.c:1064-1135)
    fused with extraction, on one host core over a bounded sample of the same stream."""
    import oracle as orc
    n, k = CPU_SAMPLE_BASES, cfg["k"]
    words = orc.synth_words(seed, n)
    t0 = time.perf_counter()
    wk, _ = orc.generate_kmers_contains(words, n, k, cfg["pattern"])
    dt = time.perf_counter() - t0
    return {"value": (n - k + 1) / dt, "unit": "rows/s", "cores": 1, "kind": "port",
            "sample": f"first {n} bases of the same synthetic stream, k={k}, pattern {cfg['pattern']}: per-row "
                      f"decode + contains() per base (dna.c:1091-1135), {len(wk)} matches, {dt:.1f} s",
            "host_cores_available": os.cpu_count()}


_CPU_PIECE = """
import sys, time
sys.path.insert(0, sys.argv[1])
import oracle as orc
seed, n, k = int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
words = orc.synth_words(seed, n)
t0 = time.perf_counter()
orc.count_kmers(words, n, k, faithful=True)
print(time.perf_counter() - t0)
"""


def cpu_baseline(k, seed):
    """The oracle timed on this box's host: one core, like the reference's one PostgreSQL backend
    (generate_kmers is not PARALLEL SAFE, dna--1.0.sql:188-191), plus -- SURVEY.md 8(d) -- the same
    work on the box's CPU share with one independent piece per core (child processes with a hard
    timeout; no merge of the pieces' groups, so it flatters the CPU).  The oracle is only the timed
    CPU baseline here; nothing of the GPU result comes from it."""
    import subprocess
    import oracle as orc
    n = CPU_SAMPLE_BASES
    words = orc.synth_words(seed, n)
    t0 = time.perf_counter()
    keys, counts = orc.count_kmers(words, n, k, faithful=True)
    dt = time.perf_counter() - t0
    out = {"value": (n - k + 1) / dt, "unit": "k-mers/s", "cores": 1, "kind": "port",
           "sample": f"first {n} bases of the same synthetic stream, k={k}: per-base decode + kmer_make "
                     f"re-encode (dna.c:803-825) + hash aggregate on kmer_hash/kmer_eq, {dt:.1f} s",
           "host_cores_available": os.cpu_count()}
    try:
        cores = max(1, min(16, os.cpu_count() or 1))       # a one-GPU box's CPU share
        per = (n // cores) // 32 * 32
        t0 = time.perf_counter()
        procs = [subprocess.Popen([sys.executable, "-c", _CPU_PIECE, ROOT, str(seed + c * (per // 32)), str(per), str(k)],
                                  stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True) for c in range(cores)]
        inner = []
        deadline = time.perf_counter() + 90.0               # for all the pieces together
        for pr in procs:
            o, _ = pr.communicate(timeout=max(1.0, deadline - time.perf_counter()))
            inner.append(float(o.strip()))
        wall = time.perf_counter() - t0
        busy = max(inner)                                   # the slowest piece: interpreter start-up excluded
        out["all_cores"] = {"value": cores * (per - k + 1) / busy, "unit": "k-mers/s", "cores": cores,
                            "sample": f"{cores} independent pieces of {per} bases, one child process per core, "
                                      f"slowest piece {busy:.1f} s ({wall:.1f} s wall with process start-up)"}
    except Exception as e:                                  # the one-core figure is the contract; this is extra
        for pr in locals().get("procs", []):
            if pr.poll() is None:
                pr.kill()
        out["all_cores"] = {"error": repr(e)[:200]}
    return out


if __name__ == "__main__":
    main()
