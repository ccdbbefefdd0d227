"""N>1 path (sharded.py): world_size-2/3 gloo runs.  CPU: the oracle stands in for the GPU engine
and checks the host logic; GPU: the product engine, all ranks on the box's one GPU."""
import importlib
import json
import os
import subprocess
import sys

import pytest

from __graft_entry__ import ROOT, load_package


def free_port():
    """a port nobody listens on right now (two test sessions on one box must not share a rendezvous port)"""
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def run_world(engine, world, n_bases, k, tmp_path, port, mode="gather", backend="gloo", parts=3):
    port = free_port()                      # (the callers' fixed numbers are kept only as labels)
    out = tmp_path / f"res_{engine}_{mode}_{world}_{n_bases}_{k}_{backend}_{parts}.json"
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0", SHARD_BACKEND=backend, SHARD_PARTS=str(parts))
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "_sharded_worker.py"),
                                       engine, str(n_bases), str(k), str(out), mode], env=env))
    for p in procs:
        assert p.wait(timeout=600) == 0
    return json.loads(out.read_text())


def test_shard_ranges_cover_exactly_once():
    pkg = load_package()
    sh = importlib.import_module(pkg.__name__ + ".shard_math")
    for n, k, w in [(1000, 31, 2), (1000, 31, 8), (64, 32, 3), (31, 31, 4), (30, 31, 2), (3_000_000_000, 31, 8)]:
        r = sh.shard_ranges(n, k, w)
        n_kmers = max(n - k + 1, 0)
        assert sum(x[1] for x in r) == n_kmers
        pos = 0
        for first, cnt, lo, hi in r:
            assert first == pos and lo % 32 == 0
            if cnt:
                assert hi == min(first + cnt + k - 1, n)      # k-1 base halo, clipped at the end
            pos += cnt


def test_word_chunks_cover_the_sequence():
    pkg = load_package()
    sh = importlib.import_module(pkg.__name__ + ".shard_math")
    for n, w in [(1000, 2), (1000, 8), (64, 3), (31, 4), (3_000_000_000, 8), (33, 5)]:
        per, chunks = sh.word_chunks(n, w)
        assert sum(c[1] for c in chunks) == n
        pos = 0
        for lo, nb in chunks:
            assert nb == 0 or lo * 32 == pos
            assert nb <= per * 32
            pos += nb


@pytest.mark.parametrize("mode", ["gather", "keys"])
@pytest.mark.parametrize("world,n_bases,k", [(2, 200_000, 31), (3, 100_001, 21), (2, 5000, 8)])
def test_sharded_count_gloo_oracle_engine(tmp_path, world, n_bases, k, mode):
    res = run_world("oracle", world, n_bases, k, tmp_path, 29511 + world + (10 if mode == "keys" else 0), mode)
    assert res["ok"] and res["sorted"], res


def test_bucket_owner_ranges_cover_exactly_once():
    pkg = load_package()
    sh = importlib.import_module(pkg.__name__ + ".shard_math")
    for nb, w in [(64, 8), (56, 8), (7, 2), (7, 3), (1, 4), (128, 6), (3, 8)]:
        r = sh.bucket_owner_ranges(nb, w)
        assert r[0][0] == 0 and r[-1][1] == nb
        for (lo, hi), (lo2, _) in zip(r, r[1:]):
            assert lo <= hi == lo2
        for b in range(nb):
            o = (b * w) // nb
            assert r[o][0] <= b < r[o][1]


def test_bucket_owner_ranges_weighted():
    pkg = load_package()
    sh = importlib.import_module(pkg.__name__ + ".shard_math")
    import random
    rnd = random.Random(5)
    for nb, w in [(68, 8), (64, 8), (7, 3), (1, 4), (128, 6), (3, 8), (68, 1)]:
        for trial in range(20):
            weights = [rnd.randint(0, 1000) if rnd.random() < 0.9 else 0 for _ in range(nb)]
            r = sh.bucket_owner_ranges_weighted(weights, w)
            assert len(r) == w and r[0][0] == 0 and r[-1][1] == nb
            for (lo, hi), (lo2, _) in zip(r, r[1:]):
                assert lo <= hi == lo2
            tot = sum(weights)
            if tot and w > 1 and nb >= 4 * w:
                share = [sum(weights[lo:hi]) for lo, hi in r]
                assert max(share) <= tot / w + max(weights), (nb, w, share)     # within one bucket of the even share
    assert sh.bucket_owner_ranges_weighted([0] * 10, 3) == sh.bucket_owner_ranges(10, 3)


@pytest.mark.parametrize("world,n_bases,k,parts", [(2, 200_000, 31, 3), (3, 100_001, 25, 3), (2, 5000, 23, 2), (3, 100_001, 25, 1),
                                                   (2, 200_000, 31, 1)])
def test_sharded_records_gloo_oracle_engine(tmp_path, world, n_bases, k, parts):
    """the record exchange's host logic (bucket owners, bucket groups, split sizes, piece boundaries) with the oracle
    standing in: one all-to-all (parts = 1) and the pipelined point-to-point rounds (parts > 1)"""
    res = run_world("oracle", world, n_bases, k, tmp_path, 0, "records", parts=parts)
    assert res["ok"] and res["sorted"], res


@pytest.mark.gpu
@pytest.mark.parametrize("world,n_bases,k,parts", [(2, 3_000_000, 31, 3), (3, 1_000_003, 27, 1), (2, 70_000, 23, 2), (1, 6_000_000, 31, 1),
                                                   (2, 3_000_000, 31, 1)])
def test_sharded_records_gloo_gpu_engine(tmp_path, world, n_bases, k, parts):
    """count_sharded_exchange_records with the product engine: every rank cuts the records of its own rows
    (dnagpu_sk_records), the buckets travel to their owners, the owners count them (dnagpu_count_records)"""
    res = run_world("gpu", world, n_bases, k, tmp_path, 0, "records", parts=parts)
    assert res["ok"] and res["sorted"], res


@pytest.mark.gpu
def test_sharded_records_rccl_one_rank(tmp_path):
    for parts in (1, 3):
        res = run_world("gpu", 1, 2_000_000, 31, tmp_path, 0, "records", backend="nccl", parts=parts)
        assert res["ok"] and res["sorted"], res


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["gather", "keys"])
@pytest.mark.parametrize("world,n_bases,k", [(2, 3_000_000, 31), (3, 1_000_003, 21), (2, 70_000, 4)])
def test_sharded_count_gloo_gpu_engine(tmp_path, world, n_bases, k, mode):
    if mode == "keys" and k < 6:
        pytest.skip("the key-exchange variant needs 2k > 10 bits")
    res = run_world("gpu", world, n_bases, k, tmp_path, 29531 + world + (10 if mode == "keys" else 0), mode)
    assert res["ok"] and res["sorted"], res


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["gather", "keys"])
def test_sharded_count_rccl_one_rank(tmp_path, mode):
    """The product backend (nccl = RCCL) with the collectives forced at world size 1: device tensors, the
    library's stream and torch's stream meet the way they do on the 8-GPU node (a one-GPU box cannot
    host two RCCL ranks)."""
    res = run_world("gpu", 1, 2_000_000, 31, tmp_path, 29561 + (1 if mode == "keys" else 0), mode, backend="nccl")
    assert res["ok"] and res["sorted"], res


@pytest.mark.gpu
def test_pool_buffer_on_another_stream(tmp_path):
    """include/dnagpu.h's stream rule for pooled buffers, exercised from a non-default torch stream (own process: torch
    brings its own ROCm runtime)"""
    out = tmp_path / "stream.json"
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "_stream_worker.py"), str(out)],
                       env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0"), timeout=600)
    assert p.returncode == 0
    res = json.loads(out.read_text())
    assert res["ok"], res
