"""Multi-GPU k-mer count: one process per GPU, contiguous sequence shards, one key exchange.

The sequence shards by contiguous chunk: rank r owns the k-mers that START in its
range of positions, and reads k-1 halo bases past its range so none is lost or counted twice.  The
count itself needs one real exchange: equal k-mers found on different ranks must meet.  Keys are
partitioned by owner = contiguous ranges of their top bits (dnagpu_partition_kmers), exchanged with
one all-to-all (RCCL over xGMI when the backend is "nccl"), and counted where they land
(dnagpu_count_keys).  The global result is the concatenation of the ranks' results in rank order,
keys ascending.

The engine is injected so that the host logic (shard arithmetic, split sizes, the collective) is
covered by world_size-2 gloo tests on CPU with the oracle standing in for the GPU; the product
engine is GpuEngine below and nothing else.
"""
import ctypes as C

import torch
import torch.distributed as dist


OWNER_BITS = 10     # dnagpu_partition_kmers partitions on the top min(2k, 10) key bits


def owner_key_range(k, owner, world):
    """[key_min, key_max] of the keys owner `owner` receives: digits d with (d * world) >> bits == owner."""
    bits = min(2 * k, OWNER_BITS)
    R = 1 << bits
    d_lo = (owner * R + world - 1) // world
    d_hi = ((owner + 1) * R + world - 1) // world       # exclusive
    shift = 2 * k - bits
    if d_hi <= d_lo:
        return 0, 0
    return d_lo << shift, (d_hi << shift) - 1


def shard_ranges(n_bases, k, world):
    """Position ranges [lo, hi) of owned k-mer starts per rank, cut on 32-base (word) boundaries.
    Returns a list of (first_kmer, n_kmers, base_lo, base_hi) where [base_lo, base_hi) are the
    bases the rank must hold (its range plus the k-1 base halo)."""
    n_kmers = n_bases - k + 1 if n_bases >= k else 0
    words = (n_kmers + 31) // 32
    per = (words + world - 1) // world
    out = []
    for r in range(world):
        lo = min(r * per * 32, n_kmers)
        hi = min((r + 1) * per * 32, n_kmers)
        base_lo = lo if hi > lo else (lo // 32) * 32      # empty shard: any word-aligned start
        base_hi = min(hi + k - 1, n_bases) if hi > lo else base_lo
        out.append((lo, hi - lo, base_lo, base_hi))
    return out


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

    def make_shard(self, seed, base_lo, base_hi):
        # word w of the global synthetic stream is splitmix64(seed + w): a shard is the same
        # generator started at its first word
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

    def empty(self, n):
        return torch.empty(max(n, 1), dtype=torch.int64, device=self.device)[:n]

    def count_keys(self, keys_t, k, key_min, key_max):
        torch.cuda.synchronize(self.device)
        return self.ctx.count_keys_device_in_range(C.c_void_p(keys_t.data_ptr()), keys_t.numel(), k,
                                                   key_min, key_max)

    def free_dna(self, dna):
        dna.free()


def exchange(send, offsets, world, engine):
    """All-to-all of the owner groups.  send: int64 tensor grouped by owner; offsets[o]..offsets[o+1]
    is owner o's group.  Returns the keys this rank owns (unordered)."""
    in_splits = [offsets[o + 1] - offsets[o] for o in range(world)]
    if world == 1:
        return send
    # gloo has no device all-to-all: stage through the host (test rigs only; RCCL moves device
    # buffers directly over xGMI)
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


def count_sharded(engine, seed, n_bases, k, rank, world, dna=None):
    """One full sharded count.  Returns (hist, dna) where hist is this rank's part of the global
    histogram (an engine object with .distinct/.total) and dna the shard (kept resident)."""
    first, n_mine, base_lo, base_hi = shard_ranges(n_bases, k, world)[rank]
    if dna is None:
        dna = engine.make_shard(seed, base_lo, base_hi)
    send, offsets = engine.partition(dna, k, n_mine, world)
    recv = exchange(send, offsets, world, engine)
    key_min, key_max = owner_key_range(k, rank, world)
    hist = engine.count_keys(recv, k, key_min, key_max)
    engine.release()
    return hist, dna
