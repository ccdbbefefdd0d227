"""Worker of the N>1 tests: one rank of a sharded count (sharded.py), gloo rendezvous on 127.0.0.1.
engine=oracle : the CPU oracle stands in for the GPU (tests the host logic: shards, halo, owner split
                sizes, the all-to-all) -- runs anywhere
engine=gpu    : the product engine on cuda:0 (every rank shares the one GPU of the box; the
                exchange is staged through the host because gloo has no device all-to-all)
Rank 0 writes {"ok": bool, ...} as JSON to the path in argv."""
import importlib
import json
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import oracle as orc  # noqa: E402
from __graft_entry__ import load_package  # noqa: E402


class OracleHist:
    def __init__(self, keys, counts):
        self.keys, self.counts = keys, counts
        self.distinct, self.total = len(keys), int(counts.sum()) if len(counts) else 0

    def download(self):
        return self.keys, self.counts

    def free(self):
        pass


class OracleEngine:
    """Same interface as sharded.GpuEngine, computed by the oracle (tests only)."""

    # ---- all-gather variant
    def make_chunk(self, seed, word_lo, n_bases_chunk):
        return (orc.synth_words(seed + word_lo, n_bases_chunk), n_bases_chunk)

    def chunk_tensor(self, chunk, per_words):
        words, n = chunk
        t = torch.zeros(per_words, dtype=torch.int64)
        t[:len(words)] = torch.from_numpy(words.view(np.int64).copy())
        return t

    def count_owned(self, words_t, n_bases, k, rank, world):
        words = words_t.numpy().view(np.uint64)[: (n_bases + 31) // 32]
        keys = orc.generate_kmers(words, n_bases, k, faithful=False)
        bits = min(2 * k, 10)
        owner = ((keys >> np.uint64(2 * k - bits)) * np.uint64(world)) >> np.uint64(bits)
        k_, c_ = orc.count_keys(keys[owner == rank])
        return OracleHist(k_, c_)

    # ---- key-exchange variant

    def make_shard(self, seed, base_lo, base_hi):
        n = base_hi - base_lo
        return (orc.synth_words(seed + base_lo // 32, n), n)

    def partition(self, dna, k, count, world):
        words, n = dna
        keys = orc.generate_kmers(words, n, k, 0, count, faithful=False)
        bits = min(2 * k, 10)
        owner = ((keys >> np.uint64(2 * k - bits)) * np.uint64(world)) >> np.uint64(bits)
        order = np.argsort(owner, kind="stable")
        offs = [int(x) for x in np.searchsorted(owner[order], np.arange(world + 1))]
        return torch.from_numpy(keys[order].view(np.int64).copy()), offs

    def release(self):
        pass

    def empty(self, n):
        return torch.empty(n, dtype=torch.int64)

    def count_keys(self, keys_t, k, key_min, key_max):
        keys = keys_t.numpy().view(np.uint64)
        assert keys.size == 0 or (int(keys.min()) >= key_min and int(keys.max()) <= key_max)
        k_, c_ = orc.count_keys(keys)
        return OracleHist(k_, c_)

    def free_dna(self, dna):
        pass

    # ---- record-exchange variant: a "record" here is (key, 1) and the bucket a hash of the key -- the host logic under
    # test (owner ranges, split sizes, piece boundaries) sees 16-byte records grouped by bucket, as from the GPU engine
    N_BUCKETS = 7

    def sk_buckets(self, global_rows, k):
        return self.N_BUCKETS

    def sk_records(self, dna, k, count, global_rows):
        words, n = dna
        keys = orc.generate_kmers(words, n, k, 0, count, faithful=False)
        bucket = ((keys * np.uint64(0x9E3779B97F4A7C15)) >> np.uint64(40)) % np.uint64(self.N_BUCKETS)
        order = np.argsort(bucket, kind="stable")
        offs = [int(x) for x in np.searchsorted(bucket[order], np.arange(self.N_BUCKETS + 1))]
        recs = np.empty(2 * len(keys), dtype=np.uint64)
        recs[0::2] = keys[order]
        recs[1::2] = 1
        return torch.from_numpy(recs.view(np.int64).copy()), offs

    def release_records(self):
        pass

    def count_records(self, recv_t, pieces, k, global_rows, last=True):
        recs = recv_t.numpy().view(np.uint64)
        assert sum(n for _, n, _ in pieces) * 2 == recs.size
        keys = recs[0::2]
        for off, n, b in pieces:                   # every piece holds records of the bucket it is said to hold
            kk = keys[off:off + n]
            assert np.all(((kk * np.uint64(0x9E3779B97F4A7C15)) >> np.uint64(40)) % np.uint64(self.N_BUCKETS) == b)
        k_, c_ = orc.count_keys(keys)
        return OracleHist(k_, c_)


def main():
    engine_name, n_bases, k, out_path = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
    mode = sys.argv[5] if len(sys.argv) > 5 else "gather"
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    backend = os.environ.get("SHARD_BACKEND", "gloo")     # "nccl" = RCCL: the product backend (GPU tests)
    if backend == "nccl":
        torch.cuda.set_device(0)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", 0))
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    pkg = load_package()
    sh = importlib.import_module(pkg.__name__ + ".sharded")
    seed = 0xD2A0003
    ctx = None
    if engine_name == "gpu":
        ctx = pkg.Context(0)
        engine = sh.GpuEngine(pkg, ctx, torch.device("cuda", 0))
    else:
        engine = OracleEngine()
    if mode == "gather":
        hist, dna = sh.count_sharded(engine, seed, n_bases, k, rank, world, always_collective=True)
    elif mode == "records":
        hist, dna = sh.count_sharded_exchange_records(engine, seed, n_bases, k, rank, world, always_collective=True,
                                                      parts=int(os.environ.get("SHARD_PARTS", "3")))
    else:
        hist, dna = sh.count_sharded_exchange_keys(engine, seed, n_bases, k, rank, world, always_collective=True)
    keys, counts = hist.download()
    parts = [None] * world
    dist.all_gather_object(parts, (np.asarray(keys), np.asarray(counts)))
    if rank == 0:
        gk = np.concatenate([p[0] for p in parts])
        gc = np.concatenate([p[1] for p in parts])
        if mode == "records":                      # unordered by construction: the ranks' groups are disjoint, in no key order
            order = np.argsort(gk, kind="stable")
            gk, gc = gk[order], gc[order]
        words = orc.synth_words(seed, n_bases)
        ok, oc = orc.count_kmers(words, n_bases, k)
        res = {"ok": bool(np.array_equal(gk, ok) and np.array_equal(gc, oc)),
               "distinct": int(len(gk)), "oracle_distinct": int(len(ok)),
               "sorted": bool(np.all(gk[1:] > gk[:-1])) if len(gk) > 1 else True,
               "per_rank": [int(len(p[0])) for p in parts], "total": int(hist.total)}
        with open(out_path, "w") as f:
            json.dump(res, f)
    hist.free()
    if engine_name == "gpu":
        dna.free()
    dist.barrier()
    dist.destroy_process_group()
    if ctx is not None:
        ctx.close()


if __name__ == "__main__":
    main()
