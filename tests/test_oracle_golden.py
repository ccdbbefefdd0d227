"""Pins the CPU oracle against the reference's own expected outputs (test.sql / README.md) and the
values SURVEY.md recorded from the reference's compiled code.  CPU only."""
import numpy as np
import pytest

import oracle as orc


def _kmers_text(dna, k, faithful=True):
    words, n = orc.dna_encode(dna)
    return [orc.kmer_decode(b, k) for b in orc.generate_kmers(words, n, k, faithful=faithful)]


# ---------------------------------------------------------------- reference's own tests

def test_generate_kmers_rows(ref_vectors):
    for v in ref_vectors["generate_kmers"]:
        assert _kmers_text(v["dna"], v["k"]) == v["rows"]
        assert _kmers_text(v["dna"], v["k"], faithful=False) == v["rows"]


def test_equals_filter(ref_vectors):
    for v in ref_vectors["equals_filter"]:
        words, n = orc.dna_encode(v["dna"])
        qlen, qbits = orc.kmer_encode(v["kmer"])
        keys, _ = orc.generate_kmers_equals(words, n, v["k"], qlen, qbits)
        assert [orc.kmer_decode(b, v["k"]) for b in keys] == v["rows"]


def test_starts_with_filter(ref_vectors):
    for v in ref_vectors["starts_with_filter"]:
        words, n = orc.dna_encode(v["dna"])
        plen, pbits = orc.kmer_encode(v["prefix"])
        keys, _ = orc.generate_kmers_starts_with(words, n, v["k"], plen, pbits)
        assert [orc.kmer_decode(b, v["k"]) for b in keys] == v["rows"]


def test_contains_filter(ref_vectors):
    for v in ref_vectors["contains_filter"]:
        words, n = orc.dna_encode(v["dna"])
        keys, _ = orc.generate_kmers_contains(words, n, v["k"], v["pattern"])
        assert [orc.kmer_decode(b, v["k"]) for b in keys] == v["rows"]


def test_qkmer_valid(ref_vectors):
    for v in ref_vectors["qkmer_valid"]:
        orc.qkmer_validate(v["pattern"])
    with pytest.raises(orc.OracleError):
        orc.qkmer_validate("ACGZ")
    with pytest.raises(orc.OracleError):
        orc.qkmer_validate("")
    with pytest.raises(orc.OracleError):
        orc.qkmer_validate("A" * 33)


def test_count_groups(ref_vectors):
    for v in ref_vectors["count"]:
        for faithful in (True, False):
            words, n = orc.dna_encode(v["dna"])
            keys, counts = orc.count_kmers(words, n, v["k"], faithful=faithful)
            got = {orc.kmer_decode(b, v["k"]): int(c) for b, c in zip(keys, counts)}
            assert got == v["groups"]
            assert list(keys) == sorted(keys)


def test_total_distinct_unique(ref_vectors):
    for v in ref_vectors["summary"]:
        words, n = orc.dna_encode(v["dna"])
        keys, counts = orc.count_kmers(words, n, v["k"], faithful=True)
        total, distinct, unique, _ = orc.hist_summary(keys, counts)
        assert (total, distinct, unique) == (v["total"], v["distinct"], v["unique"])


def test_statistical_sanity(ref_vectors):
    # test.sql:139-154 is not reproducible (unseeded generator); the seeded synthetic stream must
    # land near the same distinct/unique counts: E[distinct] = 4^k (1 - exp(-n/4^k)).
    v = ref_vectors["statistical_only"][0]
    words = orc.synth_words(0xD2A0000, v["n_bases"])
    keys, counts = orc.count_kmers(words, v["n_bases"], v["k"])
    total, distinct, unique, _ = orc.hist_summary(keys, counts)
    assert total == v["total"]
    assert abs(distinct - v["distinct"]) < 0.005 * v["distinct"]
    assert abs(unique - v["unique"]) < 0.01 * v["unique"]


# ---------------------------------------------------------------- SURVEY.md recorded values

def test_packed_words(survey_vectors):
    for v in survey_vectors["packed_words"]:
        words, n = orc.dna_encode(v["dna"])
        assert [int(w) for w in words] == [int(x, 16) for x in v["words"]]
        assert orc.dna_decode(words, n) == v["dna"]


def test_kmer_bits(survey_vectors):
    for v in survey_vectors["kmer_bits"]:
        words, n = orc.dna_encode(v["dna"])
        want = [int(x, 16) for x in v["keys"]]
        assert [int(b) for b in orc.generate_kmers(words, n, v["k"], faithful=True)] == want
        assert [int(b) for b in orc.generate_kmers(words, n, v["k"], faithful=False)] == want
    for v in survey_vectors["kmer_bits_at"]:
        words, n = orc.dna_encode(v["dna"])
        for faithful in (True, False):
            keys = orc.generate_kmers(words, n, v["k"], faithful=faithful)
            assert len(keys) == v["n"]
            for p, x in v["at"].items():
                assert int(keys[int(p)]) == int(x, 16)


def test_histogram_bits(survey_vectors):
    for v in survey_vectors["histogram_bits"]:
        words, n = orc.dna_encode(v["dna"])
        keys, counts = orc.count_kmers(words, n, v["k"])
        assert {int(a): int(c) for a, c in zip(keys, counts)} == {int(a, 16): c for a, c in v["groups"].items()}


def test_kmer_hash_values(survey_vectors):
    assert orc.pg_hashint4(1) == survey_vectors["pg_hashint4"][0]["hash"]
    for v in survey_vectors["kmer_hash"]:
        _, bits = orc.kmer_encode(v["kmer"])
        assert orc.kmer_hash(bits) == v["hash"]


def test_contains_masks(survey_vectors):
    for v in survey_vectors["contains_mask"]:
        words, n = orc.dna_encode(v["dna"])
        _, pos = orc.generate_kmers_contains(words, n, v["k"], v["pattern"])
        if "positions" in v:
            assert [int(p) for p in pos] == v["positions"]
        else:
            assert [int(p) for p in pos] == [i for i, m in enumerate(v["match"]) if m]


def test_starts_with_positions(survey_vectors):
    for v in survey_vectors["starts_with_positions"]:
        words, n = orc.dna_encode(v["dna"])
        plen, pbits = orc.kmer_encode(v["prefix"])
        _, pos = orc.generate_kmers_starts_with(words, n, v["k"], plen, pbits)
        assert [int(p) for p in pos] == v["positions"]


def test_edge_cases(survey_vectors):
    e = survey_vectors["edge_cases"]
    words, n = orc.dna_encode("ACGTACG")
    assert len(orc.generate_kmers(words, n, 7)) == e["len_eq_k_rows"]
    assert len(orc.generate_kmers(words, n, 8)) == e["len_eq_k_minus_1_rows"]
    assert len(orc.generate_kmers(words, n, 20)) == 0        # reference underflows here; we return 0 rows
    for k in e["invalid_k"]:
        with pytest.raises(orc.OracleError) as ei:
            orc.generate_kmers(words, n, k)
        assert str(ei.value) == e["invalid_k_message"]
    for bad in e["dna_bad_inputs"]:
        with pytest.raises(orc.OracleError):
            orc.dna_encode(bad)
    # contains: length mismatch is an ERROR, not false (dna.c:1106-1108)
    with pytest.raises(orc.OracleError) as ei:
        orc.contains("ACG", 4, 0)
    assert "lengths do not match" in str(ei.value)
    # ^@: prefix longer than kmer is an ERROR (dna.c:854-856); 32-base prefix matches itself here
    with pytest.raises(orc.OracleError):
        orc.starts_with(3, 0, 4, 0)
    l32, b32 = orc.kmer_encode("ACGT" * 8)
    assert orc.starts_with(l32, b32, l32, b32)
    # 'X' in kmer text equals 'A' (dna.c:413)
    assert orc.kmer_encode("ACX")[1] == orc.kmer_encode("ACA")[1]


def test_faithful_equals_fast_random():
    rng = np.random.default_rng(7)
    for n in (1, 31, 32, 33, 64, 65, 1000, 4097):
        words = orc.synth_words(int(rng.integers(1 << 40)), n)
        for k in (1, 2, 5, 16, 21, 31, 32):
            a = orc.generate_kmers(words, n, k, faithful=True)
            b = orc.generate_kmers(words, n, k, faithful=False)
            assert np.array_equal(a, b)
            assert len(a) == max(n - k + 1, 0)


def test_synth_tail_bits_zero_and_repeat():
    for n in (1, 31, 32, 33, 1000):
        w = orc.synth_words(5, n)
        assert orc.dna_decode(w, n) == orc.dna_decode(w, n)  # decodes
        if n % 32:
            assert int(w[-1]) >> (2 * (n % 32)) == 0
    n, motif = 5000, 100
    w = orc.synth_words_repeat(9, n, motif)
    s = orc.dna_decode(w, n)
    base = orc.dna_decode(orc.synth_words(9, n), n)
    assert s[: n // 2] == base[: n // 2]
    assert all(s[n // 2 + i] == base[i % motif] for i in range(n - n // 2))


def test_wire_image_known_answers():
    """dna_send/dna_recv, kmer_send/kmer_recv (dna.c:244-291, 552-597): length then packed words, each in
    network byte order.  The reference holds no test or fixture for its binary I/O (and its length field
    cannot work: pq_sendint with size 8), so these vectors are written out by hand from the format."""
    w, n = orc.dna_encode("ATCG")                       # word 0xe4
    wire = orc.dna_to_wire(w, n)
    assert wire == bytes.fromhex("0000000000000004" "00000000000000e4")
    w2, n2 = orc.dna_from_wire(wire)
    assert n2 == 4 and list(w2) == [0xE4]
    # the 71-base word-straddling vector of SURVEY.md 8(a)
    seq = "ATCGTAGCGTACGTTAGCCATGGATCCAAGTTCGATCGGCTAACGTAGCTAGGATCCTTAAGGCCATGCAT"
    w, n = orc.dna_encode(seq)
    wire = orc.dna_to_wire(w, n)
    assert wire == bytes.fromhex("0000000000000047" "5c293d2b1787b1e4" "bc1693c6c781be4e" "00000000000012d2")
    w2, n2 = orc.dna_from_wire(wire)
    assert n2 == 71 and np.array_equal(w2, w) and orc.dna_decode(w2, n2) == seq
    # bits behind the last base are cleared on receive (palloc0 invariant, dna.c:186)
    dirty = bytearray(wire)
    dirty[-8] = 0xFF
    w3, _ = orc.dna_from_wire(bytes(dirty))
    assert np.array_equal(w3, w)
    for bad in (b"", wire[:-1], wire + b"\0", bytes(8)):
        with pytest.raises(orc.OracleError):
            orc.dna_from_wire(bad)
    assert orc.kmer_to_wire(5, 0xE4) == bytes.fromhex("00000005" "00000000000000e4")
    assert orc.kmer_from_wire(bytes.fromhex("00000020" "5c293d2b1787b1e4")) == (32, 0x5C293D2B1787B1E4)
    for bad_len in (0, 33, -1):
        with pytest.raises(orc.OracleError):
            orc.kmer_from_wire(orc.kmer_to_wire(bad_len, 1))


def test_table_of_sequences_rows_are_each_sequences_own():
    """generate_kmers over a table (test.sql:140-150: LATERAL generate_kmers(d.sequence, k) per row): the rows of the
    concatenated stream restricted to windows inside one sequence == every sequence's own rows through the text path
    (dna_in + generate_kmers), faithful and fast forms; the reference's literal of test.sql:95 among the rows."""
    seqs = ["ATCGATCGATCGATCGACG", "ACG", "", "ACGTACGTACGTAG", "T" * 40, "GATTACA"]
    w, n = orc.dna_encode("".join(seqs))
    starts = np.concatenate([[0], np.cumsum([len(s) for s in seqs])]).astype(np.uint64)
    assert n == int(starts[-1])
    for k in (1, 3, 5, 8, 14, 19, 20):
        want = [orc.generate_kmers(*orc.dna_encode(s), k) for s in seqs if len(s) >= k]
        want = np.concatenate(want) if want else np.empty(0, dtype=np.uint64)
        for faithful in (True, False):
            got = orc.generate_kmers_table(w, starts, k, faithful=faithful)
            assert np.array_equal(got, want), (k, faithful)
    keys, counts = orc.count_keys(orc.generate_kmers_table(w, starts, 5))
    groups = {orc.kmer_decode(a, 5): int(b) for a, b in zip(keys, counts)}
    assert groups["ATCGA"] == 4 and groups["TTTTT"] == 36 and groups["ACGTA"] == 3     # test.sql:95's 4 + the other rows
