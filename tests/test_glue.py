"""The host-side mirror of the reference's operator surface (glue/), exercised with the statements
of the reference's own test script.  Type I/O and per-datum operators are host logic (CPU tests);
generate_kmers / WHERE-fused / GROUP BY forms run on the GPU (gpu tests)."""
import importlib

import pytest

from __graft_entry__ import load_package


@pytest.fixture(scope="module")
def g():
    pkg = load_package()
    return importlib.import_module(pkg.__name__ + ".glue")


# ------------------------------------------------------------------ CPU: types and per-datum operators

def test_dna_io(g):
    d = g.dna("ATCG")
    assert len(d) == 4 and str(d) == "ATCG"                                  # test.sql:26-30
    s = "ATCGATCGATCGATCGATCGATCGATCGATCGATCGATCGATCGATCGATCGATCGATCGATCG"     # test.sql:32,43
    assert len(g.dna(s)) == 64 and str(g.dna(s)) == s
    for bad, msg in (("", "DNA sequence cannot be empty"), ("ATNG", "Invalid character in DNA sequence: N"),
                     ("atcg", "Invalid character in DNA sequence: a")):
        with pytest.raises(g.GlueError) as ei:
            g.dna(bad)
        assert str(ei.value) == msg                                          # dna.c:161,166


def test_binary_io(g):
    """dna_send/dna_recv, kmer_send/kmer_recv (dna.c:244-291, 552-597) against the oracle's wire image"""
    import oracle as orc
    seq = "ATCGTAGCGTACGTTAGCCATGGATCCAAGTTCGATCGGCTAACGTAGCTAGGATCCTTAAGGCCATGCAT"
    for s_ in ("A", "ATCG", seq, seq[:32], seq[:33], seq * 7):
        wire = g.dna(s_).send()
        assert wire == orc.dna_to_wire(*orc.dna_encode(s_))
        assert str(g.dna.recv(wire)) == s_
    for bad, msg in ((b"", "insufficient data left in message"), (bytes(8), "DNA sequence cannot be empty"),
                     (orc.dna_to_wire(*orc.dna_encode(seq))[:-8], "insufficient data left in message")):
        with pytest.raises(g.GlueError) as ei:
            g.dna.recv(bad)
        assert str(ei.value) == msg
    k = g.kmer("ATCGA")
    assert k.send() == orc.kmer_to_wire(5, 0xE4) and g.kmer.recv(k.send()) == k
    with pytest.raises(g.GlueError) as ei:
        g.kmer.recv(orc.kmer_to_wire(33, 1))
    assert str(ei.value) == "Invalid K-mer length: must be between 1 and 32"     # dna.c:567


def test_kmer_io_and_eq(g):
    assert str(g.kmer("ACGTAC")) == "ACGTAC"
    assert g.kmer("ATCG") == g.kmer("ATCG") and g.kmer("ATCG") != g.kmer("GTCA")
    assert g.kmer("ATCG") != g.kmer("ATCGA")                                 # same bits, other length (dna.c:658)
    assert g.kmer("ACX") == g.kmer("ACA")                                    # 'X' encodes as 'A' (dna.c:413)
    with pytest.raises(g.GlueError) as ei:
        g.kmer("A" * 33)
    assert str(ei.value) == "K-mer length cannot exceed 32 nucleotides"      # dna.c:467
    with pytest.raises(g.GlueError) as ei:
        g.kmer("ACNT")
    assert str(ei.value) == "Invalid character in K-mer sequence: 'N'"       # dna.c:473
    with pytest.raises(g.GlueError):
        g.kmer("")


def test_qkmer_io(g, ref_vectors):
    for v in ref_vectors["qkmer_valid"]:
        assert str(g.qkmer(v["pattern"])) == v["pattern"]                    # test.sql:75-84
    with pytest.raises(g.GlueError) as ei:
        g.qkmer("ACGZ")
    assert str(ei.value) == "Invalid character in qkmer pattern: Z"          # dna.c:894


def test_kmer_hash_values(g, survey_vectors):
    for v in survey_vectors["kmer_hash"]:
        assert g.kmer_hash(g.kmer(v["kmer"])) == v["hash"]


def test_scalar_operators(g):
    assert g.starts_with(g.kmer("ACT"), g.kmer("AC")) and not g.starts_with(g.kmer("CTG"), g.kmer("AC"))
    with pytest.raises(g.GlueError) as ei:
        g.starts_with(g.kmer("AC"), g.kmer("ACT"))
    assert str(ei.value) == "Prefix length cannot exceed kmer length"        # dna.c:855
    k32 = g.kmer("ACGT" * 8)
    assert g.starts_with(k32, k32)                                           # reference: UB shift (dna.c:862)
    assert g.contains(g.qkmer("DNMSRN"), g.kmer("GTACGC")) and not g.contains(g.qkmer("DNMSRN"), g.kmer("ACGTAC"))
    assert not g.contains(g.qkmer("NU"), g.kmer("AT"))                       # 'U' matches nothing (dna.c:1070)
    with pytest.raises(g.GlueError) as ei:
        g.contains(g.qkmer("ACG"), g.kmer("ACGT"))
    assert str(ei.value) == "Qkmer pattern and kmer lengths do not match"    # dna.c:1107


# ------------------------------------------------------------------ GPU: the statements of test.sql

@pytest.mark.gpu
def test_sql_generate_kmers(g, ref_vectors):
    for v in ref_vectors["generate_kmers"]:                                  # test.sql:46-58
        assert [str(k) for k in g.generate_kmers(v["dna"], v["k"])] == v["rows"]
    assert g.generate_kmers("ACGTACG", 8) == [] and len(g.generate_kmers("ACGTACG", 7)) == 1
    for k in (0, 33):
        with pytest.raises(g.GlueError) as ei:
            g.generate_kmers("ACGT", k)
        assert str(ei.value) == "Invalid k value: must be between 1 and 32"  # dna.c:773


@pytest.mark.gpu
def test_sql_where_operators(g, ref_vectors):
    for v in ref_vectors["equals_filter"]:                                   # test.sql:61-65
        assert [str(k) for k in g.generate_kmers_where(v["dna"], v["k"], "=", g.kmer(v["kmer"]))] == v["rows"]
    for v in ref_vectors["starts_with_filter"]:                              # test.sql:67-73
        assert [str(k) for k in g.generate_kmers_where(v["dna"], v["k"], "^@", g.kmer(v["prefix"]))] == v["rows"]
    for v in ref_vectors["contains_filter"]:                                 # test.sql:86-92
        assert [str(k) for k in g.generate_kmers_where(v["dna"], v["k"], "@>", g.qkmer(v["pattern"]))] == v["rows"]
    with pytest.raises(g.GlueError) as ei:
        g.generate_kmers_where("ACGTACGT", 4, "@>", g.qkmer("ACG"))
    assert str(ei.value) == "Qkmer pattern and kmer lengths do not match"
    with pytest.raises(g.GlueError) as ei:
        g.generate_kmers_where("ACGTACGT", 3, "^@", g.kmer("ACGT"))
    assert str(ei.value) == "Prefix length cannot exceed kmer length"


@pytest.mark.gpu
def test_sql_group_by_count(g, ref_vectors):
    for v in ref_vectors["count"]:                                           # test.sql:95-104
        rows, _ = g.count_kmers(v["dna"], v["k"])
        assert {str(k): c for k, c in rows} == v["groups"]
        by_count = sorted(rows, key=lambda r: -r[1])                         # ORDER BY count(*) DESC
        assert str(by_count[0][0]) == "ATCGA" and by_count[0][1] == 4
    for v in ref_vectors["summary"]:                                         # test.sql:107-119, README.md:120-134
        _, totals = g.count_kmers(v["dna"], v["k"])
        assert totals == (v["total"], v["distinct"], v["unique"])


@pytest.mark.gpu
def test_sql_group_by_count_multi_gpu(g, ref_vectors):
    """the same statements with count_kmers sharded over several ranks through dnagpu_count_multi (one process;
    the ranks share device 0 on a one-GPU box): same groups, same totals, rows in ascending key order"""
    import numpy as np
    import oracle as orc
    try:
        for n_ranks in (2, 3, 8):
            g.set_gpus([0] * n_ranks, 2)                                     # DNAGPU_MULTI_COPY
            for v in ref_vectors["count"]:                                   # test.sql:95-104
                rows, _ = g.count_kmers(v["dna"], v["k"])
                assert {str(k): c for k, c in rows} == v["groups"]
            for v in ref_vectors["summary"]:                                 # test.sql:107-119
                _, totals = g.count_kmers(v["dna"], v["k"])
                assert totals == (v["total"], v["distinct"], v["unique"])
        g.set_gpus([0, 0, 0], 2)
        n, k = 200_003, 12
        words = orc.synth_words(77, n)
        rows, totals = g.count_kmers(orc.dna_decode(words, n), k)
        ok, oc = orc.count_kmers(words, n, k)
        assert [r[0].c.bit_sequence for r in rows] == [int(x) for x in ok]
        assert [r[1] for r in rows] == [int(x) for x in oc]
        assert totals[0] == n - k + 1 and totals[1] == len(ok)
    finally:
        g.set_gpus([0])


@pytest.mark.gpu
def test_srf_windows_cross_refills(g):
    # more rows than one GPU window (4 Mi rows): rows keep position order across refills
    import numpy as np
    import oracle as orc
    n = 4_400_000
    words = orc.synth_words(12, n)
    text = orc.dna_decode(words, n)
    rows = g.generate_kmers(text, 31)
    want = orc.generate_kmers(words, n, 31, faithful=False)
    assert len(rows) == len(want)
    got = np.fromiter((r.c.bit_sequence for r in rows), dtype=np.uint64, count=len(rows))
    assert np.array_equal(got, want)
