"""Multi-GPU k-mer count: one process per GPU, the sequence sharded by contiguous chunk.

Counting needs one real exchange: equal k-mers found in different chunks must meet.  Two ways, both
with the same ownership rule (owner o of W owns the keys whose top-10-bit digit d satisfies
(d * W) >> 10 == o: contiguous, ascending key ranges, so the concatenation of the ranks' results in
rank order is the global result in ascending key order):

``count_sharded`` (default)   all-gather the PACKED SEQUENCE (2 bits per base, RCCL over xGMI), then
    every rank scans all of it and keeps only the keys it owns (dnagpu_count_kmers_owned).  The
    exchange is 64x smaller than moving keys -- at 3 Gbase each rank receives 0.75 GB instead of
    sending 21/W GB of keys -- at the price of every rank extracting every window, which is cheap
    (a few VALU ops per window).  With two ranks joined by one xGMI link the key exchange would take
    longer than the whole single-GPU count; this does not.

``count_sharded_exchange_keys``   rank r extracts only its own chunk (+ k-1 halo bases), partitions
    the keys by owner (dnagpu_partition_kmers), exchanges them with one all-to-all and counts what
    it receives (dnagpu_count_keys_in_range).  Kept as the alternative for many GPUs / very long
    sequences, where scanning everything on every rank stops being free.

The engine is injected so that the host logic (chunk arithmetic, collectives, split sizes) is
covered by world_size-2/3 gloo tests on CPU with the oracle standing in for the GPU; the product
engine is GpuEngine below and nothing else.
"""
import ctypes as C

import torch
import torch.distributed as dist

from .shard_math import (OWNER_BITS, bucket_group_cuts, bucket_owner_ranges, bucket_owner_ranges_weighted,  # noqa: F401
                         owner_key_range, shard_ranges, word_chunks)  # (pure arithmetic: no torch)


class HistParts:
    """The histogram of an owner whose buckets were counted group by group (the pipelined record exchange): the same
    read methods as one histogram, the groups of part i before those of part i + 1."""

    def __init__(self, hists):
        self.hists = list(hists)

    @property
    def distinct(self):
        return sum(h.distinct for h in self.hists)

    @property
    def total(self):
        return sum(h.total for h in self.hists)

    @property
    def is_sorted(self):
        return False

    def download(self):
        import numpy as np
        parts = [h.download() for h in self.hists]
        if not parts:
            return np.zeros(0, dtype=np.uint64), np.zeros(0, dtype=np.uint64)
        return np.concatenate([p[0] for p in parts]), np.concatenate([p[1] for p in parts])

    def summary(self):
        t = [0, 0, 0, 0]
        for h in self.hists:
            t = [(a + b) & ((1 << 64) - 1) for a, b in zip(t, h.summary())]
        return tuple(t)

    def free(self):
        for h in self.hists:
            h.free()
        self.hists = []


class _DevArray:
    """Zero-copy view of library-owned device memory for torch (``__cuda_array_interface__``)."""

    def __init__(self, ptr, n):
        self.__cuda_array_interface__ = {"shape": (n,), "typestr": "<i8", "data": (int(ptr), False),
                                         "version": 2, "strides": None}


class GpuEngine:
    """The product engine: HIP kernels through the C-ABI."""

    def __init__(self, pkg, ctx, device):
        self.pkg, self.ctx, self.device = pkg, ctx, device
        self._held = []

    # ---- resident input
    def make_chunk(self, seed, word_lo, n_bases_chunk):
        """words [word_lo, ...) of the global synthetic stream (word w = splitmix64(seed + w))"""
        return self.ctx.synth(seed + word_lo, n_bases_chunk)

    def chunk_tensor(self, dna, per_words):
        """the chunk as an int64 tensor of per_words words (zero padded) for the all-gather"""
        nw = (dna.n_bases + 31) // 32
        if nw == per_words:                 # a full chunk: the library's buffer itself, no copy
            self.ctx.synchronize()
            return torch.as_tensor(_DevArray(dna.device_words, nw), device=self.device)
        t = torch.zeros(per_words, dtype=torch.int64, device=self.device)
        if nw:
            src = torch.as_tensor(_DevArray(dna.device_words, nw), device=self.device)
            self.ctx.synchronize()
            t[:nw].copy_(src)
        return t

    def empty(self, n):
        return torch.empty(max(n, 1), dtype=torch.int64, device=self.device)[:n]

    def count_owned(self, words_t, n_bases, k, rank, world):
        torch.cuda.synchronize(self.device)
        dna = self.ctx.wrap(C.c_void_p(words_t.data_ptr()), words_t.numel(), n_bases)
        try:
            return self.ctx.count_kmers_owned(dna, k, rank, world)
        finally:
            dna.free()

    # ---- key-exchange variant
    def make_shard(self, seed, base_lo, base_hi):
        assert base_lo % 32 == 0
        return self.ctx.synth(seed + base_lo // 32, base_hi - base_lo)

    def partition(self, dna, k, count, world):
        if count == 0:
            return self.empty(0), [0] * (world + 1)
        ptr, offs = self.ctx.partition_kmers(dna, k, 0, count, world)
        self._held.append(ptr)
        t = torch.as_tensor(_DevArray(ptr, count), device=self.device)
        return t, [int(x) for x in offs]

    def release(self):
        for p in self._held:
            self.ctx.buffer_free(p)
        self._held = []

    def count_keys(self, keys_t, k, key_min, key_max):
        torch.cuda.synchronize(self.device)
        return self.ctx.count_keys_device_in_range(C.c_void_p(keys_t.data_ptr()), keys_t.numel(), k,
                                                   key_min, key_max)

    def free_dna(self, dna):
        dna.free()

    # ---- record-exchange variant
    def sk_buckets(self, global_rows, k):
        return self.ctx.sk_buckets(global_rows, k)

    def sk_records(self, dna, k, count, global_rows):
        """-> (int64 tensor view of the records, two words each, grouped by bucket; offsets per bucket in records)"""
        r = self.ctx.sk_records(dna, k, 0, count, global_rows)
        self._records = r
        self._phases0 = list(self.ctx.last_phase_times())      # (level 0; the count's phases follow in count_records)
        self._phases1 = []
        if r.n_records == 0:
            return self.empty(0), [0] * (r.n_buckets + 1)
        return torch.as_tensor(_DevArray(r.device_ptr, 2 * r.n_records), device=self.device), [int(x) for x in r.offsets]

    def phase_times(self):
        """phase times of the last sharded count: the record cut (if any) followed by the count"""
        return list(getattr(self, "_phases0", [])) + list(getattr(self, "_phases1", []))

    def count_records(self, recv_t, pieces, k, global_rows, last=True):
        """pieces: [(offset in records inside recv_t, n_records, bucket)]"""
        torch.cuda.synchronize(self.device)
        base = recv_t.data_ptr() if recv_t.numel() else 0
        h = self.ctx.count_records([(base + 16 * off, n, b) for off, n, b in pieces if n], k, global_rows)
        # (the pipelined exchange counts the buckets group by group: every group's phases, the same names again)
        self._phases1 = list(getattr(self, "_phases1", [])) + list(self.ctx.last_phase_times())
        return h

    def release_records(self):
        """the rank's own records (the send buffer of the exchange) go back to the pool"""
        if getattr(self, "_records", None) is not None:
            self._records.free()
            self._records = None


def gather_sequence(chunk_t, world, engine, always=False):
    """All-gather of the packed chunks -> the whole packed sequence on every rank.
    always: run the collective at world size 1 too (the RCCL smoke test on a one-GPU box)."""
    if world == 1 and not always:
        return chunk_t
    via_host = dist.get_backend() == "gloo" and chunk_t.is_cuda     # gloo has no device all-gather
    if via_host:
        full_h = torch.empty(world * chunk_t.numel(), dtype=torch.int64)
        dist.all_gather_into_tensor(full_h, chunk_t.cpu())
        full = engine.empty(full_h.numel())
        full.copy_(full_h)
        return full
    full = engine.empty(world * chunk_t.numel())
    dist.all_gather_into_tensor(full, chunk_t)
    return full


def count_sharded(engine, seed, n_bases, k, rank, world, chunk=None, always_collective=False):
    """One full sharded count (all-gather of the packed sequence + owner-filtered count).
    Returns (hist, chunk): this rank's part of the global histogram and its resident chunk."""
    per, chunks = word_chunks(n_bases, world)
    if chunk is None:
        chunk = engine.make_chunk(seed, *chunks[rank])
    chunk_t = engine.chunk_tensor(chunk, per)
    full = gather_sequence(chunk_t, world, engine, always_collective)
    hist = engine.count_owned(full, n_bases, k, rank, world)
    return hist, chunk


def exchange(send, offsets, world, engine, always=False):
    """All-to-all of the owner groups.  send: int64 tensor grouped by owner; offsets[o]..offsets[o+1]
    is owner o's group.  Returns the keys this rank owns (unordered)."""
    in_splits = [offsets[o + 1] - offsets[o] for o in range(world)]
    if world == 1 and not always:
        return send
    via_host = dist.get_backend() == "gloo" and send.is_cuda
    sizes = torch.tensor(in_splits, dtype=torch.int64, device="cpu" if via_host else send.device)
    recv_sizes = torch.empty_like(sizes)
    dist.all_to_all_single(recv_sizes, sizes)
    out_splits = [int(x) for x in recv_sizes.tolist()]
    if via_host:
        recv_h = torch.empty(sum(out_splits), dtype=torch.int64)
        dist.all_to_all_single(recv_h, send.cpu(), out_splits, in_splits)
        recv = engine.empty(sum(out_splits))
        recv.copy_(recv_h)
        return recv
    recv = engine.empty(sum(out_splits))
    dist.all_to_all_single(recv, send, out_splits, in_splits)
    return recv


def count_sharded_exchange_keys(engine, seed, n_bases, k, rank, world, dna=None, always_collective=False):
    """The key-exchange variant.  Returns (hist, dna) like count_sharded."""
    first, n_mine, base_lo, base_hi = shard_ranges(n_bases, k, world)[rank]
    if dna is None:
        dna = engine.make_shard(seed, base_lo, base_hi)
    send, offsets = engine.partition(dna, k, n_mine, world)
    recv = exchange(send, offsets, world, engine, always_collective)
    key_min, key_max = owner_key_range(k, rank, world)
    hist = engine.count_keys(recv, k, key_min, key_max)
    engine.release()
    return hist, dna


def count_sharded_exchange_records(engine, seed, n_bases, k, rank, world, dna=None, always_collective=False, parts=1):
    """The record-exchange variant (k >= 21): rank r cuts the super-k-mer records of its OWN rows only, every coarse
    bucket goes to its owner (1.8 bytes per k-mer at k = 31; nothing is swept twice and no rank touches the whole
    sequence), and the owner counts the records it received.  parts = 1 (default): one all-to-all, then the count.
    parts > 1: the exchange is PIPELINED with the count like dnagpu_count_multi_unordered's -- an owner's buckets travel in
    `parts` groups (point-to-point sends and receives, all posted at once, group after group), and group g is counted
    while group g + 1 is still in flight (covered by the gloo tests; never run on nccl with more than one rank).  The ranks'
    histograms are disjoint (a k-mer's bucket depends on its content alone) but in no key order.
    Returns (hist, dna) like count_sharded; hist is a HistParts when more than one group was counted."""
    first, n_mine, base_lo, base_hi = shard_ranges(n_bases, k, world)[rank]
    global_rows = max(n_bases - k + 1, 0)
    if dna is None:
        dna = engine.make_shard(seed, base_lo, base_hi)
    n_buckets = engine.sk_buckets(global_rows, k)
    send, boffs = engine.sk_records(dna, k, n_mine, global_rows)        # records grouped by bucket = by owner
    # what every rank holds of every bucket: the receiver needs the piece boundaries inside what it is sent
    via_host = dist.get_backend() == "gloo"
    counts = torch.tensor([boffs[b + 1] - boffs[b] for b in range(n_buckets)], dtype=torch.int64,
                          device="cpu" if via_host or not send.is_cuda else send.device)
    all_counts = torch.empty(world * n_buckets, dtype=torch.int64, device=counts.device)
    if world == 1 and not always_collective:
        all_counts.copy_(counts)
    else:
        dist.all_gather_into_tensor(all_counts, counts)
    all_counts = all_counts.cpu().view(world, n_buckets)
    # owners: contiguous bucket ranges, balanced by the records the buckets hold (the same split on every rank); every
    # owner's range in `parts` groups (the same cuts on every rank, too)
    weights = [int(x) for x in all_counts.sum(dim=0).tolist()]
    owners = bucket_owner_ranges_weighted(weights, world)
    if int(parts) <= 1:
        # one group: ONE all-to-all (the collective every backend has; what arrives is known from the counts), then the count
        lo, hi = owners[rank]
        in_splits = [2 * (boffs[min(owners[o][1], n_buckets)] - boffs[min(owners[o][0], n_buckets)]) for o in range(world)]
        out_splits = [2 * int(all_counts[src, lo:hi].sum()) for src in range(world)]
        if world == 1 and not always_collective:
            recv = send
        elif via_host and send.is_cuda:                                 # gloo has no device all-to-all (test rig)
            recv_h = torch.empty(sum(out_splits), dtype=torch.int64)
            dist.all_to_all_single(recv_h, send.cpu(), out_splits, in_splits)
            recv = engine.empty(sum(out_splits))
            recv.copy_(recv_h)
        else:
            recv = engine.empty(sum(out_splits))
            dist.all_to_all_single(recv, send, out_splits, in_splits)
        pieces, pos = [], 0
        for src in range(world):                                        # the all-to-all delivers source after source
            for b in range(lo, hi):
                n = int(all_counts[src, b])
                pieces.append((pos, n, b))
                pos += n
        assert 2 * pos == recv.numel(), (pos, recv.numel())
        hist = engine.count_records(recv, pieces, k, global_rows)
        engine.release_records()
        return hist, dna
    gcuts = [bucket_group_cuts(weights, lo_o, hi_o, parts) for lo_o, hi_o in owners]
    n_groups = max(1, int(parts))
    staged = via_host and send.is_cuda                                  # gloo has no device transport (test rig)
    send_x = send.cpu() if staged else send
    recv_bufs, reqs_of = [], []
    for g in range(n_groups):
        g_lo, g_hi = gcuts[rank][g], gcuts[rank][g + 1]
        sizes = [2 * int(all_counts[src, g_lo:g_hi].sum()) for src in range(world)]
        buf = torch.empty(max(sum(sizes), 1), dtype=torch.int64, device=send_x.device)[:sum(sizes)]
        ops, at = [], 0
        for q in range(world):
            src = (rank + q) % world                                    # (own piece first)
            seg = buf[sum(sizes[:src]):sum(sizes[:src]) + sizes[src]]
            if src == rank:
                o_lo, o_hi = gcuts[rank][g], gcuts[rank][g + 1]
                if sizes[src]:
                    seg.copy_(send_x[2 * boffs[o_lo]:2 * boffs[o_hi]])
            elif sizes[src]:
                ops.append(dist.P2POp(dist.irecv, seg, src))
        for q in range(1, world):
            dst = (rank + q) % world
            d_lo, d_hi = gcuts[dst][g], gcuts[dst][g + 1]
            if boffs[d_hi] > boffs[d_lo]:
                ops.append(dist.P2POp(dist.isend, send_x[2 * boffs[d_lo]:2 * boffs[d_hi]], dst))
        reqs_of.append(dist.batch_isend_irecv(ops) if ops else [])
        recv_bufs.append((buf, sizes, g_lo, g_hi))
    hists = []
    for g in range(n_groups):
        for r in reqs_of[g]:
            r.wait()
        buf, sizes, g_lo, g_hi = recv_bufs[g]
        if buf.numel() == 0:
            continue
        if staged:
            dev = engine.empty(buf.numel())
            dev.copy_(buf)
            buf = dev
        pieces, pos = [], 0
        for src in range(world):                                        # the pieces lie source after source
            for b in range(g_lo, g_hi):
                n = int(all_counts[src, b])
                pieces.append((pos, n, b))
                pos += n
        assert 2 * pos == buf.numel(), (pos, buf.numel())
        hists.append(engine.count_records(buf, pieces, k, global_rows, last=(g == n_groups - 1)))
    engine.release_records()
    if not hists:
        hists.append(engine.count_records(engine.empty(0), [], k, global_rows, last=True))
    return (hists[0] if len(hists) == 1 else HistParts(hists)), dna
