"""GPU parity: the HIP back-end (through the C-ABI, libdnagpu.so) against the CPU oracle on the same
seeded inputs, against the reference's golden vectors, and -- at BASELINE.json sizes -- through
size-independent properties.  Bit-exact: everything on this path is integer work."""
import os

import numpy as np
import pytest

import oracle as orc
from __graft_entry__ import load_package

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def pkg():
    return load_package()


@pytest.fixture(scope="module")
def ctx(pkg):
    c = pkg.Context(0)
    yield c
    c.close()


def assert_same(got, want, what):
    got = np.asarray(got)
    want = np.asarray(want)
    assert got.shape == want.shape, f"{what}: {got.shape[0]} values, oracle has {want.shape[0]}"
    if not np.array_equal(got, want):
        bad = np.flatnonzero(got != want)
        i = int(bad[0])
        raise AssertionError(f"{what}: {bad.size} of {got.size} differ; first at {i}: "
                             f"got {int(got[i]):#x}, oracle {int(want[i]):#x}")


def check_hist(hist, ok, oc, what):
    gk, gc = hist.download()
    assert hist.distinct == len(ok), f"{what}: {hist.distinct} groups, oracle {len(ok)}"
    assert_same(gk, ok, what + " keys")
    assert_same(gc, oc, what + " counts")
    assert hist.summary() == orc.hist_summary(ok, oc), what + " summary"


# ------------------------------------------------------------------ input generator

@pytest.mark.parametrize("n", [1, 31, 32, 33, 64, 1000, 100_003])
def test_synth_matches_oracle(ctx, n):
    d = ctx.synth(0xABC, n)
    assert_same(d.download(), orc.synth_words(0xABC, n), f"synth n={n}")
    d.free()
    d = ctx.synth(0xABC, n, motif_len=37)
    assert_same(d.download(), orc.synth_words_repeat(0xABC, n, 37), f"synth repeat n={n}")
    d.free()


# ------------------------------------------------------------------ text <-> packed (dna_in / dna_out / kmer_out)

@pytest.mark.parametrize("n", [1, 4, 31, 32, 33, 63, 64, 65, 1000, 100_003])
def test_pack_unpack_matches_oracle(ctx, n):
    words = orc.synth_words(555 + n, n)
    text = orc.dna_decode(words, n)
    d = ctx.pack(text)
    assert d.n_bases == n
    assert_same(d.download(), words, f"pack n={n}")
    assert ctx.unpack(d) == text
    if n > 40:
        assert ctx.unpack(d, 7, 33) == text[7:40]
    d.free()


@pytest.mark.parametrize("n", [1, 31, 32, 33, 64, 65, 1000, 100_003, 2_000_017])
def test_wire_image_matches_oracle(ctx, pkg, n):
    """dna_send / dna_recv on the device (dna.c:244-291): byte for byte the oracle's wire image"""
    words = orc.synth_words(777 + n, n)
    wire = orc.dna_to_wire(words, n)
    d = ctx.upload(words, n)
    assert ctx.to_wire(d) == wire, f"to_wire n={n}"
    d.free()
    r = ctx.from_wire(wire)
    assert r.n_bases == n
    assert_same(r.download(), words, f"from_wire n={n}")
    r.free()
    if n % 32:                                  # bits behind the last base are cleared (dna.c:186)
        dirty = bytearray(wire)
        dirty[-8] |= 0xC0
        r = ctx.from_wire(bytes(dirty))
        assert_same(r.download(), words, f"from_wire dirty tail n={n}")
        r.free()


def test_wire_image_errors(ctx, pkg):
    wire = orc.dna_to_wire(*orc.dna_encode("ATCGATCG"))
    for bad, code in ((wire[:-1], 5), (wire + b"\0" * 8, 5), (b"\0" * 4, 5), (bytes(8), 11)):
        with pytest.raises(pkg.DnaGpuError) as ei:
            ctx.from_wire(bad)
        assert ei.value.code == code            # DNAGPU_ERR_BAD_ARG / DNAGPU_ERR_DNA_EMPTY


def test_pack_errors_are_the_references(ctx, pkg):
    with pytest.raises(pkg.DnaGpuError) as ei:
        ctx.pack("")
    assert ei.value.message == "DNA sequence cannot be empty"                      # dna.c:161
    good = "ACGT" * 5000
    for pos, ch in ((0, "N"), (17, "a"), (19_999, "X"), (4097, "U")):
        bad = good[:pos] + ch + good[pos + 1:]
        with pytest.raises(pkg.DnaGpuError) as ei:
            ctx.pack(bad)
        assert ei.value.message == f"Invalid character in DNA sequence: {ch}"      # dna.c:166
        assert ei.value.bad_pos == pos
    # two bad characters: the first one is reported, as the reference's left-to-right scan does
    bad = good[:100] + "z" + good[101:9000] + "y" + good[9001:]
    with pytest.raises(pkg.DnaGpuError) as ei:
        ctx.pack(bad)
    assert ei.value.message.endswith(": z") and ei.value.bad_pos == 100


def test_kmers_to_text(ctx, ref_vectors):
    v = ref_vectors["generate_kmers"][0]
    d = ctx.pack(v["dna"])
    assert ctx.kmers_to_text(ctx.generate_kmers(d, v["k"]), v["k"]) == v["rows"]     # test.sql:46-58
    d.free()
    rng = np.random.default_rng(5)
    for k in (1, 7, 31, 32):
        keys = rng.integers(0, 1 << 62, size=5000, dtype=np.uint64)
        if k < 32:
            keys &= np.uint64((1 << (2 * k)) - 1)
        assert ctx.kmers_to_text(keys, k) == [orc.kmer_decode(int(x), k) for x in keys]


# ------------------------------------------------------------------ generate_kmers

def test_generate_kmers_reference_rows(ctx, ref_vectors):
    for v in ref_vectors["generate_kmers"]:
        words, n = orc.dna_encode(v["dna"])
        d = ctx.upload(words, n)
        rows = [orc.kmer_decode(b, v["k"]) for b in ctx.generate_kmers(d, v["k"])]
        assert rows == v["rows"]
        d.free()


def test_generate_kmers_survey_bits(ctx, survey_vectors):
    for v in survey_vectors["kmer_bits"]:
        words, n = orc.dna_encode(v["dna"])
        d = ctx.upload(words, n)
        assert [int(b) for b in ctx.generate_kmers(d, v["k"])] == [int(x, 16) for x in v["keys"]]
        d.free()
    for v in survey_vectors["kmer_bits_at"]:
        words, n = orc.dna_encode(v["dna"])
        d = ctx.upload(words, n)
        keys = ctx.generate_kmers(d, v["k"])
        assert len(keys) == v["n"]
        for p, x in v["at"].items():
            assert int(keys[int(p)]) == int(x, 16)
        d.free()


@pytest.mark.parametrize("n", [1, 5, 31, 32, 33, 63, 64, 65, 4095, 4096, 4097, 70_001])
def test_generate_kmers_random(ctx, n):
    words = orc.synth_words(1000 + n, n)
    d = ctx.upload(words, n)
    for k in (1, 2, 5, 15, 16, 17, 21, 31, 32):
        want = orc.generate_kmers(words, n, k, faithful=False)
        assert_same(ctx.generate_kmers(d, k), want, f"generate_kmers n={n} k={k}")
    d.free()


def test_generate_kmers_faithful_oracle_and_windows(ctx):
    n = 20_011
    words = orc.synth_words(77, n)
    d = ctx.upload(words, n)
    for k in (5, 21, 31, 32):
        want = orc.generate_kmers(words, n, k, faithful=True)     # the reference's per-base loops
        assert_same(ctx.generate_kmers(d, k), want, f"faithful k={k}")
        for first, count in ((0, 1), (1, 4097), (31, 33), (len(want) - 1, 1), (len(want), 0), (4999, 10_001)):
            assert_same(ctx.generate_kmers(d, k, first, count), want[first:first + count],
                        f"window k={k} first={first} count={count}")
    d.free()


def test_generate_kmers_edge_cases(ctx, pkg):
    words, n = orc.dna_encode("ACGTACG")
    d = ctx.upload(words, n)
    assert len(ctx.generate_kmers(d, 7)) == 1          # len == k
    assert len(ctx.generate_kmers(d, 8)) == 0          # len == k-1
    assert len(ctx.generate_kmers(d, 20)) == 0         # the reference underflows here; 0 rows
    for k in (0, 33, -1):
        with pytest.raises(pkg.DnaGpuError) as ei:
            ctx.generate_kmers(d, k)
        assert ei.value.message == "Invalid k value: must be between 1 and 32"
    with pytest.raises(pkg.DnaGpuError):
        ctx.generate_kmers(d, 3, first=4, count=5)     # outside the row range
    d.free()


# ------------------------------------------------------------------ fused WHERE operators

def test_filters_reference_rows(ctx, pkg, ref_vectors):
    for v in ref_vectors["equals_filter"]:
        words, n = orc.dna_encode(v["dna"])
        d = ctx.upload(words, n)
        ln, bits = orc.kmer_encode(v["kmer"])
        keys, _, tot = ctx.generate_kmers_filtered(d, v["k"], pkg.Filter.equals(ln, bits))
        assert [orc.kmer_decode(b, v["k"]) for b in keys] == v["rows"] and tot == len(v["rows"])
        d.free()
    for v in ref_vectors["starts_with_filter"]:
        words, n = orc.dna_encode(v["dna"])
        d = ctx.upload(words, n)
        ln, bits = orc.kmer_encode(v["prefix"])
        keys, _, _ = ctx.generate_kmers_filtered(d, v["k"], pkg.Filter.starts_with(ln, bits))
        assert [orc.kmer_decode(b, v["k"]) for b in keys] == v["rows"]
        d.free()
    for v in ref_vectors["contains_filter"]:
        words, n = orc.dna_encode(v["dna"])
        d = ctx.upload(words, n)
        keys, _, _ = ctx.generate_kmers_filtered(d, v["k"], pkg.Filter.contains(v["pattern"]))
        assert [orc.kmer_decode(b, v["k"]) for b in keys] == v["rows"]
        d.free()


def test_filters_survey_positions(ctx, pkg, survey_vectors):
    for v in survey_vectors["contains_mask"]:
        words, n = orc.dna_encode(v["dna"])
        d = ctx.upload(words, n)
        _, pos, _ = ctx.generate_kmers_filtered(d, v["k"], pkg.Filter.contains(v["pattern"]))
        want = v["positions"] if "positions" in v else [i for i, m in enumerate(v["match"]) if m]
        assert [int(p) for p in pos] == want
        d.free()


PATTERNS_21 = ["NNNNNNNNNNWSNNNNNNNNN", "RYNNNNNNNNNNNNNNNNNNN", "ATCGUWSMKRYBDHVNNNNNN"[:21],
               "BDHVBDHVBDHVBDHVBDHVB", "NNNNNNNNNNNNNNNNNNNNN", "ANNNNNNNNNNNNNNNNNNNT"]


@pytest.mark.parametrize("n", [21, 4096 + 20, 250_007])
def test_contains_random(ctx, pkg, n):
    words = orc.synth_words(4242, n)
    d = ctx.upload(words, n)
    for pat in PATTERNS_21:
        wk, wp = orc.generate_kmers_contains(words, n, 21, pat)
        gk, gp, tot = ctx.generate_kmers_filtered(d, 21, pkg.Filter.contains(pat))
        assert tot == len(wk), f"{pat}: {tot} matches, oracle {len(wk)}"
        assert_same(gk, wk, f"contains {pat} keys")
        assert_same(gp, wp, f"contains {pat} positions")
    # every IUPAC letter at k=1 and a 32-long pattern
    for ch in "ATCGUWSMKRYBDHVN":
        wk, wp = orc.generate_kmers_contains(words, n, 1, ch)
        gk, gp, _ = ctx.generate_kmers_filtered(d, 1, pkg.Filter.contains(ch))
        assert_same(gp, wp, f"contains '{ch}' positions")
    if n >= 32:
        pat = "NWSNNNNNNNNNNNNNNNNNNNNNNNNNNNRY"
        assert len(pat) == 32
        wk, wp = orc.generate_kmers_contains(words, n, 32, pat)
        gk, gp, _ = ctx.generate_kmers_filtered(d, 32, pkg.Filter.contains(pat))
        assert_same(gk, wk, "contains k=32 keys")
        assert_same(gp, wp, "contains k=32 positions")
    d.free()


def test_filters_random_patterns_windows_alignment(ctx, pkg):
    """random IUPAC patterns (every letter, both halves of a 32-base pattern), random k, unaligned windows,
    device outputs at either 16-byte parity, keys-only / positions-only"""
    import ctypes as C
    rng = np.random.default_rng(20261004)
    n = 70_000
    words = orc.synth_words(31337, n)
    d = ctx.upload(words, n)
    letters = "ATCGWSMKRYBDHVN"
    for trial in range(60):
        k = int(rng.integers(1, 33))
        # mostly N with a few constrained positions, so that something matches
        pat = ["N"] * k
        for _ in range(int(rng.integers(1, 5))):
            pat[int(rng.integers(0, k))] = letters[int(rng.integers(0, len(letters)))]
        pat = "".join(pat)
        total = n - k + 1
        first = int(rng.integers(0, total // 2))
        count = int(rng.integers(0, total - first + 1))
        wk, wp = orc.generate_kmers_contains(words, n, k, pat)
        sel = (wp >= first) & (wp < first + count)
        gk, gp, tot = ctx.generate_kmers_filtered(d, k, pkg.Filter.contains(pat), first=first, count=count)
        assert tot == int(sel.sum()), f"{pat} k={k} first={first} count={count}: {tot} vs {int(sel.sum())}"
        assert_same(gk, wk[sel], f"{pat} k={k} window keys")
        assert_same(gp, wp[sel], f"{pat} k={k} window positions")
    # device outputs: both parities of the 16-byte alignment, keys only, positions only, small cap
    k, pat = 21, "NNNNNNNNNNWSNNNNNNNNN"
    wk, wp = orc.generate_kmers_contains(words, n, k, pat)
    nk = n - k + 1
    for koff, poff in ((0, 0), (8, 8), (8, 0), (0, 8)):
        kb, pb = ctx.buffer_alloc(nk * 8 + 16), ctx.buffer_alloc(nk * 8 + 16)
        for want_k, want_p, cap in ((True, True, nk), (True, False, nk), (False, True, nk), (True, True, 1001)):
            m = ctx.count_matches_device(d, k, pkg.Filter.contains(pat), 0, nk,
                                         C.c_void_p(kb + koff) if want_k else None,
                                         C.c_void_p(pb + poff) if want_p else None, cap)
            assert m == len(wk)
            w = min(cap, m)
            if want_k:
                assert_same(ctx.download_u64(kb + koff, w), wk[:w], f"device keys off {koff}/{poff} cap {cap}")
            if want_p:
                assert_same(ctx.download_u64(pb + poff, w), wp[:w], f"device positions off {koff}/{poff} cap {cap}")
        ctx.buffer_free(kb)
        ctx.buffer_free(pb)
    # cap = 0 / no outputs: the count alone
    assert ctx.count_matches_device(d, k, pkg.Filter.contains(pat), 0, nk, None, None, 0) == len(wk)
    d.free()


@pytest.mark.parametrize("n", [8192 * 8192 + 77, 40_000_003])
def test_contains_many_groups(ctx, pkg, n):
    """more tiles than workgroups of one sweep: several tiles per group, running offsets across tiles"""
    words = orc.synth_words(0xF00D, n)
    d = ctx.upload(words, n)
    for pat in ("NNNNNNNNNNWSNNNNNNNNN", "ACNNNNNNNNNNNNNNNNNNG"):
        wk, wp = orc.generate_kmers_contains(words, n, 21, pat)
        gk, gp, tot = ctx.generate_kmers_filtered(d, 21, pkg.Filter.contains(pat))
        assert tot == len(wk)
        assert_same(gk, wk, f"{pat} keys n={n}")
        assert_same(gp, wp, f"{pat} positions n={n}")
    d.free()


def test_starts_with_and_equals_random(ctx, pkg):
    n = 300_001
    words = orc.synth_words(99, n)
    d = ctx.upload(words, n)
    for k, prefix in ((3, "AC"), (21, "ACG"), (21, "G"), (31, "TTTT"), (32, "CA"), (5, "ACGTA")):
        ln, bits = orc.kmer_encode(prefix)
        wk, wp = orc.generate_kmers_starts_with(words, n, k, ln, bits)
        gk, gp, tot = ctx.generate_kmers_filtered(d, k, pkg.Filter.starts_with(ln, bits))
        assert tot == len(wk)
        assert_same(gk, wk, f"^@ {prefix} k={k} keys")
        assert_same(gp, wp, f"^@ {prefix} k={k} positions")
    # equality: pick a k-mer that occurs, at k=8 it repeats
    keys8 = orc.generate_kmers(words, n, 8, faithful=False)
    q = int(keys8[1234])
    wk, wp = orc.generate_kmers_equals(words, n, 8, 8, q)
    gk, gp, tot = ctx.generate_kmers_filtered(d, 8, pkg.Filter.equals(8, q))
    assert tot == len(wk) and tot >= 1
    assert_same(gp, wp, "= positions")
    # a kmer of another length never equals (kmer_eq compares lengths, dna.c:658)
    _, _, tot = ctx.generate_kmers_filtered(d, 8, pkg.Filter.equals(7, q & 0x3FFF))
    assert tot == 0
    # window + cap smaller than the number of matches: first `cap` rows, total still reported
    ln, bits = orc.kmer_encode("A")
    wk, wp = orc.generate_kmers_starts_with(words, n, 21, ln, bits)
    gk, gp, tot = ctx.generate_kmers_filtered(d, 21, pkg.Filter.starts_with(ln, bits), cap=1000)
    assert tot == len(wk) and len(gk) == 1000
    assert_same(gk, wk[:1000], "cap keys")
    sel = (wp >= 5000) & (wp < 5000 + 123_456)
    gk, gp, tot = ctx.generate_kmers_filtered(d, 21, pkg.Filter.starts_with(ln, bits), first=5000, count=123_456)
    assert_same(gp, wp[sel], "window positions")
    d.free()


def test_filter_errors(ctx, pkg):
    words, n = orc.dna_encode("ACGTACGTAC")
    d = ctx.upload(words, n)
    with pytest.raises(pkg.DnaGpuError) as ei:
        ctx.generate_kmers_filtered(d, 4, pkg.Filter.contains("ACG"))
    assert ei.value.message == "Qkmer pattern and kmer lengths do not match"
    with pytest.raises(pkg.DnaGpuError) as ei:
        ctx.generate_kmers_filtered(d, 3, pkg.Filter.starts_with(4, 0))
    assert ei.value.message == "Prefix length cannot exceed kmer length"
    with pytest.raises(pkg.DnaGpuError):
        ctx.generate_kmers_filtered(d, 3, pkg.Filter.contains("AZG"))
    with pytest.raises(pkg.DnaGpuError):
        ctx.generate_kmers_filtered(d, 3, pkg.Filter.contains(""))
    # no rows -> the operator is never evaluated -> no error (as in the reference)
    _, _, tot = ctx.generate_kmers_filtered(d, 11, pkg.Filter.contains("ACG"))
    assert tot == 0
    # 32-base prefix: compares all 64 bits (the reference's shift by 64 is undefined behaviour)
    w32, n32 = orc.dna_encode("ACGT" * 8 + "A")
    d32 = ctx.upload(w32, n32)
    ln, bits = orc.kmer_encode("ACGT" * 8)
    _, pos, tot = ctx.generate_kmers_filtered(d32, 32, pkg.Filter.starts_with(ln, bits))
    assert tot == 1 and int(pos[0]) == 0
    d.free()
    d32.free()


# ------------------------------------------------------------------ GROUP BY count(*)

def test_count_reference_groups(ctx, ref_vectors, survey_vectors):
    for v in ref_vectors["count"]:
        words, n = orc.dna_encode(v["dna"])
        d = ctx.upload(words, n)
        h = ctx.count_kmers(d, v["k"])
        gk, gc = h.download()
        assert {orc.kmer_decode(a, v["k"]): int(c) for a, c in zip(gk, gc)} == v["groups"]
        h.free()
        d.free()
    for v in ref_vectors["summary"]:
        words, n = orc.dna_encode(v["dna"])
        d = ctx.upload(words, n)
        h = ctx.count_kmers(d, v["k"])
        total, distinct, unique, _ = h.summary()
        assert (total, distinct, unique) == (v["total"], v["distinct"], v["unique"])
        h.free()
        d.free()
    for v in survey_vectors["histogram_bits"]:
        words, n = orc.dna_encode(v["dna"])
        d = ctx.upload(words, n)
        h = ctx.count_kmers(d, v["k"])
        gk, gc = h.download()
        assert {int(a): int(c) for a, c in zip(gk, gc)} == {int(a, 16): c for a, c in v["groups"].items()}
        h.free()
        d.free()


# 6143-6145: the leaf capacity; 1_000_003: leaves around 3900 keys (both leaf classes in one list);
# 1_400_000: leaves around 5470 keys (the six-keys-per-thread class only)
COUNT_SIZES = [1, 2, 100, 4095, 4096, 4097, 6143, 6144, 6145, 8193, 50_000, 1_000_003, 1_400_000]


@pytest.mark.parametrize("n", COUNT_SIZES)
def test_count_random(ctx, n):
    words = orc.synth_words(31337 + n, n)
    d = ctx.upload(words, n)
    for k in (1, 2, 3, 5, 6, 8, 11, 13, 16, 21, 31, 32):
        if n < k:
            h = ctx.count_kmers(d, k)
            assert h.distinct == 0 and h.total == 0
            h.free()
            continue
        ok, oc = orc.count_kmers(words, n, k)
        h = ctx.count_kmers(d, k)
        check_hist(h, ok, oc, f"count n={n} k={k}")
        h.free()
    d.free()


@pytest.mark.parametrize("n,ks", [(300_007, (1, 4, 7, 8)), (5_000_011, (2, 5, 6, 7, 8)), (17_000_003, (9,))])
def test_count_dense_short_kmers(ctx, n, ks):
    """2k <= 18 bits and enough rows: the dense table path (no tree).  Whole sequence, a window, a
    repeat-rich input and poly-A (every window on one counter)."""
    words = orc.synth_words(4242 + n, n)
    d = ctx.upload(words, n)
    for k in ks:
        ok, oc = orc.count_kmers(words, n, k)
        h = ctx.count_kmers(d, k)
        check_hist(h, ok, oc, f"dense n={n} k={k}")
        assert h.total == n - k + 1
        h.free()
    k = ks[-1]
    first, cnt = 12345, n - 100_000
    keys = orc.generate_kmers(words, n, k, first, cnt, faithful=False)
    ok, oc = orc.count_keys(keys)
    h = ctx.count_kmers(d, k, first, cnt)
    check_hist(h, ok, oc, f"dense window n={n} k={k}")
    h.free()
    d.free()
    if n < 10_000_000:
        for words2 in (orc.synth_words_repeat(7, n, 1000), orc.dna_encode("A" * n)[0]):
            d2 = ctx.upload(words2, n)
            ok, oc = orc.count_kmers(words2, n, k)
            h = ctx.count_kmers(d2, k)
            check_hist(h, ok, oc, f"dense skewed n={n} k={k}")
            h.free()
            d2.free()


def test_count_faithful_oracle(ctx):
    # the oracle's faithful path (per-base decode + re-encode + hash aggregate), config-1 shaped
    n = 200_000
    words = orc.synth_words(5, n)
    d = ctx.upload(words, n)
    for k in (5, 21, 31):
        ok, oc = orc.count_kmers(words, n, k, faithful=True)
        h = ctx.count_kmers(d, k)
        check_hist(h, ok, oc, f"faithful count k={k}")
        h.free()
    d.free()


@pytest.mark.parametrize("n,motif", [(100_000, 1000), (3_000_000, 1000), (2_000_000, 7), (500_000, 250_000)])
def test_count_repeat_rich(ctx, n, motif):
    words = orc.synth_words_repeat(4, n, motif)
    d = ctx.synth(4, n, motif_len=motif)
    assert_same(d.download(), words, "repeat input")
    for k in (5, 12, 21, 31, 32):
        ok, oc = orc.count_kmers(words, n, k)
        h = ctx.count_kmers(d, k)
        check_hist(h, ok, oc, f"repeat-rich n={n} motif={motif} k={k}")
        h.free()
    d.free()


@pytest.mark.parametrize("seq", ["A" * 100_000, "G" * 70_001, "AT" * 60_000, "ACGT" * 50_000,
                                 "A" * 5000 + "C" * 5000 + "G" * 5000 + "T" * 5000])
def test_count_low_complexity(ctx, seq):
    words, n = orc.dna_encode(seq)
    d = ctx.upload(words, n)
    for k in (1, 4, 16, 31, 32):
        ok, oc = orc.count_kmers(words, n, k)
        h = ctx.count_kmers(d, k)
        check_hist(h, ok, oc, f"low complexity {seq[:8]}..x{n} k={k}")
        h.free()
    d.free()


def _heavy_hitter_words(kind, n):
    """random sequences with heavy k-mers planted word-aligned (numpy only: no reference involved)"""
    nw = (n + 31) // 32
    rng = np.random.default_rng(len(kind) * 1000 + n)
    w = rng.integers(0, 2**63, nw, dtype=np.uint64) * np.uint64(2) + rng.integers(0, 2, nw, dtype=np.uint64)
    if kind == "motif":                       # a 64-base motif 9000 times: 34 k-mers (k=31) just over a leaf each
        step = (nw // 9000) & ~1
        w[0:step * 9000:step] = np.uint64(0x1234567890ABCDEF)
        w[1:step * 9000:step] = np.uint64(0x0FEDCBA987654321)
    elif kind == "half-polyA":                # one k-mer with half of all rows, sharing its top digits with others
        w[: nw // 2] = 0
    elif kind == "two-heavy":                 # two heavy k-mers that differ in their lowest base only
        w[: nw // 3] = 0
        w[nw // 3: 2 * nw // 3] = np.uint64(0x0000000100000001)
    elif kind == "tandem":                    # a 224-base tandem repeat: 224 k-mers, each with 1/224 of the rows
        w = np.tile(w[:7], nw // 7 + 1)[:nw].copy()
    r = n % 32
    if r:
        w[-1] &= np.uint64((1 << (2 * r)) - 1)
    return w


@pytest.mark.parametrize("kind,n", [("motif", 12_000_000), ("half-polyA", 3_000_001), ("two-heavy", 2_500_000),
                                    ("tandem", 4_000_000)])
def test_count_heavy_hitters(ctx, kind, n):
    """nodes that stay in place (constant, or common prefix) and deep full-width splits: count_kernels.hip
    level_hist/level_children; the oracle is the checker"""
    words = _heavy_hitter_words(kind, n)
    d = ctx.upload(words, n)
    for k in ((31,) if kind == "motif" else (12, 21, 31, 32)):
        ok, oc = orc.count_kmers(words, n, k)
        h = ctx.count_kmers(d, k)
        check_hist(h, ok, oc, f"heavy hitters {kind} n={n} k={k}")
        h.free()
    d.free()


def test_count_windows_and_linearity(ctx):
    n, k = 700_000, 21
    words = orc.synth_words_repeat(8, n, 5000)
    d = ctx.upload(words, n)
    keys = orc.generate_kmers(words, n, k, faithful=False)
    for first, count in ((0, 1), (17, 4096), (1000, 300_000), (len(keys) - 5, 5), (123, 0)):
        ok, oc = orc.count_keys(keys[first:first + count])
        h = ctx.count_kmers(d, k, first, count)
        check_hist(h, ok, oc, f"count window first={first} count={count}")
        h.free()
    # linearity: the histogram of the whole equals the merge of the histograms of two halves
    half = len(keys) // 2
    ha = ctx.count_kmers(d, k, 0, half)
    hb = ctx.count_kmers(d, k, half, len(keys) - half)
    hw = ctx.count_kmers(d, k)
    ak, ac = ha.download()
    bk, bc = hb.download()
    merged = {}
    for kk, cc in zip(np.concatenate([ak, bk]), np.concatenate([ac, bc])):
        merged[int(kk)] = merged.get(int(kk), 0) + int(cc)
    wk, wc = hw.download()
    assert merged == {int(a): int(b) for a, b in zip(wk, wc)}
    for h in (ha, hb, hw):
        h.free()
    d.free()


def test_count_keys_device(ctx):
    # dnagpu_count_keys over an arbitrary key array produced on the device
    n, k = 600_000, 31
    d = ctx.synth(21, n)
    words = d.download()
    nk = n - k + 1
    buf = ctx.buffer_alloc(nk * 8)
    ctx.generate_kmers_device(d, k, 0, nk, buf)
    h = ctx.count_keys_device(buf, nk, k)
    ok, oc = orc.count_kmers(words, n, k)
    check_hist(h, ok, oc, "count_keys")
    h.free()
    ctx.buffer_free(buf)
    d.free()


@pytest.mark.parametrize("heavy,copies,light", [(1, 5000, 0), (1, 5000, 477), (1, 6144, 0), (5, 1000, 1100),
                                                (40, 100, 1500), (1, 3000, 3000), (2, 2500, 1), (1, 900, 100)])
def test_count_keys_crafted_leaves(ctx, heavy, copies, light):
    """single leaves with long bins (count_kernels.hip leaves_kernel: pivots, padding slots, the three leaf
    classes): `heavy` keys of `copies` copies + `light` random keys, 64 such leaves side by side"""
    rng = np.random.default_rng(heavy * 7919 + copies + light)
    L, per, k = 64, heavy * copies + light, 31
    ids = np.repeat(np.arange(L, dtype=np.uint64), per) << np.uint64(56)
    pay = rng.integers(0, 2**56, L * per, dtype=np.uint64)
    hv = rng.integers(0, 2**56, heavy, dtype=np.uint64)
    hv[0] |= np.uint64(0xFFF << 44)                  # (one heavy key in the leaf's last bins)
    pay.reshape(L, per)[:, :heavy * copies] = np.repeat(hv, copies)
    keys = ids | pay
    rng.shuffle(keys)
    ok, oc = np.unique(keys, return_counts=True)
    d = ctx.upload(keys, 32 * len(keys))             # the key array as the words of a packed sequence: device memory
    h = ctx.count_keys_device(d.device_words, len(keys), k)      # (used as scratch by the count)
    check_hist(h, ok, oc.astype(np.uint64), f"crafted leaves heavy={heavy}x{copies} light={light}")
    h.free()
    d.free()


# ------------------------------------------------------------------ GROUP BY without a promised order (super-k-mer engine)

def check_hist_unordered(hist, ok, oc, what):
    """same multiset of (key, count) groups as the oracle; order of the groups free (PostgreSQL's is unspecified)"""
    gk, gc = hist.download()
    assert hist.distinct == len(ok), f"{what}: {hist.distinct} groups, oracle {len(ok)}"
    order = np.argsort(gk, kind="stable")
    assert_same(gk[order], ok, what + " keys")
    assert_same(gc[order], oc, what + " counts")
    assert hist.summary() == orc.hist_summary(ok, oc), what + " summary"
    # the raw device arrays (dnagpu_hist_device_keys / _counts over dnagpu_hist_extent slots): the slots whose count is
    # not 0 are exactly the groups (an unordered histogram pads the range of a bucket that held copies)
    ext = hist.extent
    assert ext >= hist.distinct
    if 0 < ext <= 50_000_000 and hist.n_parts == 1:      # (a histogram of several parts has no arrays of its own)
        rk = hist.ctx.download_u64(hist.device_keys, ext)
        rc = hist.ctx.download_u64(hist.device_counts, (ext + 1) // 2).view(np.uint32)[:ext]
        keep = rc != 0
        assert int(keep.sum()) == hist.distinct, f"{what}: {int(keep.sum())} non-padding slots, {hist.distinct} groups"
        o2 = np.argsort(rk[keep], kind="stable")
        assert_same(rk[keep][o2], ok, what + " raw keys")
        assert_same(rc[keep][o2].astype(np.uint64), oc, what + " raw counts")


@pytest.mark.parametrize("n,k,first", [(5_000_011, 31, 0), (4_500_000, 32, 0), (6_000_000, 27, 7), (5_000_000, 23, 33),
                                       (9_000_000, 31, 12345), (17_000_029, 31, 0), (40_000_000, 29, 1),
                                       (3_000_000, 24, 0), (3_000_000, 25, 5), (3_000_000, 26, 0), (3_000_000, 28, 31),
                                       (3_000_000, 30, 0), (70_000, 31, 3), (1_000, 32, 0),
                                       (5_000_011, 21, 0), (4_000_000, 22, 9), (40_000_000, 21, 1), (1_000, 21, 0), (70_000, 22, 3),
                                       (5_000_011, 20, 0), (40_000_000, 20, 7), (1_000, 20, 0)])
def test_count_unordered_superkmers(ctx, pkg, n, k, first):
    """dnagpu_count_kmers_unordered on sequences long enough for super-k-mer partitioning: the groups are the
    oracle's (sorted on the host for the comparison); a window that does not start on a word boundary"""
    words = orc.synth_words(0x5EED + n, n)
    d = ctx.upload(words, n)
    keys = orc.generate_kmers(words, n, k, faithful=False)[first:]
    ok, oc = orc.count_keys(keys)
    ctx.set_debug(pkg.DEBUG_FORCE_SUPERKMER)         # also for the k below 29, where the tree is the faster engine
    try:
        h = ctx.count_kmers_unordered(d, k, first=first)
    finally:
        ctx.set_debug(0)
    assert not h.is_sorted
    assert h.total == len(keys)
    check_hist_unordered(h, ok, oc, f"unordered n={n} k={k} first={first}")
    h.free()
    rows = len(keys)                                 # the default choice of engine (dnagpu_api.hip: sk_is_default)
    if rows >= (1 << 25) and (k >= 23 or (k == 22 and rows <= (1 << 31)) or (k == 21 and rows <= (1 << 29)) or
                              (k == 20 and rows <= (1 << 28))):
        h = ctx.count_kmers_unordered(d, k, first=first)
        assert not h.is_sorted
        assert h.summary() == orc.hist_summary(ok, oc)
        h.free()
    # short sequences and short k-mers take the ordinary engine behind the same entry point
    cnt12 = min(1_000_000, n - 12 + 1 - first)
    h = ctx.count_kmers_unordered(d, 12, first=first, count=cnt12)
    assert h.is_sorted
    ok, oc = orc.count_keys(orc.generate_kmers(words, n, 12, faithful=False)[first:first + cnt12])
    check_hist(h, ok, oc, "unordered entry, short k")
    h.free()
    d.free()


@pytest.mark.parametrize("n,k,kind", [(3_000_000, 31, "random"), (3_000_001, 25, "planted"), (20_000_000, 27, "random"),
                                      (6_000_000, 21, "planted")])
def test_level1_speculative_and_exact_agree(ctx, pkg, n, k, kind):
    """level 1 of the record engine without its histogram (the default: mid buckets are regions sized from their parent),
    the exact level (DNAGPU_DEBUG_NO_SPEC1) and the fall-back after an overflow -- forced (DNAGPU_DEBUG_SPEC1_OVERFLOW) and
    real: copies of one 64-base segment planted all over a random sequence (coarse buckets stay even, so the speculative
    sweep runs, and the segment's mid buckets overflow their regions) -- all give the oracle's groups"""
    words = orc.synth_words(0xC0FFEE + n, n)
    if kind == "planted":
        rng = np.random.default_rng(n)
        seg = words[1000:1002].copy()
        for p in rng.integers(0, len(words) - 3, size=4000):
            words[p:p + 2] = seg
    d = ctx.upload(words, n)
    ok, oc = orc.count_keys(orc.generate_kmers(words, n, k, faithful=False))
    ctx.set_profiling(True)
    for name, flag in (("speculative", 0), ("exact", pkg.DEBUG_NO_SPEC1), ("overflow forced", pkg.DEBUG_SPEC1_OVERFLOW)):
        ctx.set_debug(pkg.DEBUG_FORCE_SUPERKMER | flag)
        try:
            h = ctx.count_kmers_unordered(d, k)
        finally:
            ctx.set_debug(0)
        phases = {a for a, _ in ctx.last_phase_times()}
        assert not h.is_sorted
        check_hist_unordered(h, ok, oc, f"level 1 {name}: n={n} k={k} {kind}")
        h.free()
        # which way level 1 went: the speculative sweep ("sk_spec1") alone on random sequence, followed by the exact level
        # ("sk_hist1") where the planted copies -- or the test flag -- overflowed a region; the exact level alone when asked
        if name == "exact":
            assert "sk_spec1" not in phases and "sk_hist1" in phases, phases
        else:
            assert "sk_spec1" in phases, phases
            assert ("sk_hist1" in phases) == (kind == "planted" or name == "overflow forced"), (name, kind, phases)
    ctx.set_profiling(False)
    d.free()


@pytest.mark.parametrize("n,k,motif", [(40_000_000, 31, 1000), (20_000_000, 27, 64), (40_000_000, 31, 100_000), (12_000_000, 21, 300)])
def test_level1_sampled_regions_keep_repeats_speculative(ctx, pkg, n, k, motif):
    """a motif tiled over the second half of a long sequence makes the coarse buckets uneven (its records go to the few buckets
    of its minimizers).  Round 3 sent such inputs through the exact level 1 (a histogram sweep over all records); now the
    regions of the speculative sweep come from a histogram over an eighth of the records ("sk_sample1"): no "sk_hist1"
    phase, the oracle's groups.  (Sequences an oracle can count here have too few coarse buckets to show the unevenness:
    DNAGPU_DEBUG_SAMPLE1 takes the sampled regions regardless; bench.py --motif at 3 Gbase takes them by itself.)"""
    words = orc.synth_words_repeat(0x5A3B1E + n, n, motif)
    d = ctx.upload(words, n)
    ok, oc = orc.count_keys(orc.generate_kmers(words, n, k, faithful=False))
    ctx.set_profiling(True)
    ctx.set_debug(pkg.DEBUG_FORCE_SUPERKMER | pkg.DEBUG_SAMPLE1)
    try:
        h = ctx.count_kmers_unordered(d, k)
    finally:
        ctx.set_debug(0)
    phases = {a for a, _ in ctx.last_phase_times()}
    check_hist_unordered(h, ok, oc, f"sampled regions: n={n} k={k} motif={motif}")
    h.free()
    assert "sk_sample1" in phases and "sk_spec1" in phases and "sk_hist1" not in phases, phases
    ctx.set_profiling(False)
    d.free()


@pytest.mark.parametrize("n,k,kind", [(3_000_000, 31, "random"), (5_000_001, 25, "sample misses"), (20_000_000, 27, "random"),
                                      (6_000_000, 21, "motif"), (70_000, 31, "random")])
def test_level0_slabs_and_exact_agree(ctx, pkg, n, k, kind):
    """level 0 of the record engine without its histogram sweep (chunks reserve slabs sized from a sampled histogram; the
    default from 2^29 rows, forced here), the exact pair (DNAGPU_DEBUG_NO_SLAB0) and the fall-back -- forced
    (DNAGPU_DEBUG_SLAB0_OVERFLOW) and real (the sampled windows are poly-A, which makes few records, the rest is random:
    the slabs are far too small and the chunks run out of slots) -- all give the oracle's groups"""
    if kind == "motif":
        words = orc.synth_words_repeat(0xC0DE + n, n, 500)
    else:
        words = orc.synth_words(0xC0DE + n, n)
    if kind == "sample misses":                      # the sample: four histogram tiles (4 x 8064 rows) every 64 x that many rows
        span = 4 * 8064
        for lo in range(0, n, 64 * span):
            words[lo // 32:(lo + span) // 32 + 2] = np.uint64(0)
    d = ctx.upload(words, n)
    ok, oc = orc.count_keys(orc.generate_kmers(words, n, k, faithful=False))
    ctx.set_profiling(True)
    for name, flag in (("slabs", pkg.DEBUG_SLAB0), ("exact", pkg.DEBUG_NO_SLAB0), ("overflow forced", pkg.DEBUG_SLAB0_OVERFLOW)):
        ctx.set_debug(pkg.DEBUG_FORCE_SUPERKMER | flag)
        try:
            h = ctx.count_kmers_unordered(d, k)
        finally:
            ctx.set_debug(0)
        phases = {a for a, _ in ctx.last_phase_times()}
        assert not h.is_sorted
        check_hist_unordered(h, ok, oc, f"level 0 {name}: n={n} k={k} {kind}")
        h.free()
        # which way level 0 went: the sample + slab sweep alone on random sequence; followed by the exact pair ("sk_hist0")
        # where the chunks ran out of slots or the test flag says so; the exact pair alone when asked
        if name == "exact":
            assert "sk_sample0" not in phases and "sk_hist0" in phases, phases
        else:
            assert "sk_sample0" in phases, phases
            if kind == "random" and name == "slabs":
                assert "sk_hist0" not in phases, phases
            if name == "overflow forced" or kind == "sample misses":
                assert "sk_hist0" in phases, (name, kind, phases)
    ctx.set_profiling(False)
    d.free()


def genome_like_words(n, seed=0x6E0):
    """tools/genome_like.py: an Alu-like family of diverged copies, exact segmental duplications, microsatellites"""
    import sys
    tools = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools")
    if tools not in sys.path:
        sys.path.insert(0, tools)
    import genome_like
    return genome_like.packed_words(n, seed)


@pytest.mark.parametrize("kind", ["motif1000", "motif37", "motif100000", "polyA", "half-polyA", "quarter-polyA", "AT",
                                  "small-polyA", "small-AT", "genome-like"])
@pytest.mark.parametrize("k", [31, 21])
def test_count_unordered_repeats(ctx, pkg, kind, k):
    """repeat-rich and low-complexity inputs through the unordered entry: heavy buckets are split further by the
    ordinary levels; a bucket too heavy for one workgroup sends the whole count to the ordinary engine (k = 31: 15-base
    minimizers; k = 21: 13-base ones)"""
    n = 6_000_000
    if kind.startswith("motif"):
        words = orc.synth_words_repeat(91, n, int(kind[5:]))
    elif kind == "polyA":
        words = np.zeros((n + 31) // 32, dtype=np.uint64)
    elif kind == "AT":
        words, _ = orc.dna_encode("AT" * (n // 2))
    elif kind in ("small-polyA", "small-AT"):
        # 3 % of the sequence is one repeat: its bucket is heavy (forced engine: > 3 x the mean) but small enough to
        # leave the record path alone, through the eight-wave expansion, while the rest stays on it
        words = orc.synth_words(94, n)
        rep = 0 if kind == "small-polyA" else int(orc.dna_encode("AT" * 16)[0][0])
        words[1000:1000 + 200_000 // 32] = rep
    elif kind == "quarter-polyA":
        words = orc.synth_words(93, n)
        words[: len(words) // 4] = 0
    elif kind == "genome-like":
        # near-copies: a minimizer's bucket holds thousands of k-mers that differ in a base or two -- the expansion's
        # tree splits them as key_mix(key) and turns the groups back (sk_unmix)
        words = genome_like_words(n)
    else:
        words = orc.synth_words(92, n)
        words[: len(words) // 2] = 0
    d = ctx.upload(words, n)
    ok, oc = orc.count_kmers(words, n, k)
    h = ctx.count_kmers_unordered(d, k)
    check_hist_unordered(h, ok, oc, f"unordered {kind}")
    h.free()
    # the engine forced: its "heavy bucket" limit is then 16 K k-mers, so the repeats' buckets leave the record path
    # through the eight-wave expansion (what buckets of millions of k-mers do at full size)
    # (heavy mid buckets: split by d2 with the chunked level kernels, then sk_count_big -- and, with
    # DEBUG_HEAVY_EXPAND, the older path: expanded to keys as a whole)
    for flags, what in ((pkg.DEBUG_FORCE_SUPERKMER, "heavy buckets split by the level kernels"),
                        (pkg.DEBUG_FORCE_SUPERKMER | pkg.DEBUG_HEAVY_EXPAND, "heavy buckets expanded")):
        ctx.set_debug(flags)
        try:
            h = ctx.count_kmers_unordered(d, k)
            check_hist_unordered(h, ok, oc, f"unordered {kind}, {what}")
            h.free()
        finally:
            ctx.set_debug(0)
    # a short sequence (few, small buckets) of the same kind, the engine forced
    ctx.set_debug(pkg.DEBUG_FORCE_SUPERKMER)
    try:
        m = 300_000
        ok, oc = orc.count_kmers(words, m, k)
        ds = ctx.upload(words[: (m + 31) // 32].copy() & np.uint64(0xFFFFFFFFFFFFFFFF), m) if m % 32 == 0 else None
        if ds is not None:
            h = ctx.count_kmers_unordered(ds, k)
            check_hist_unordered(h, ok, oc, f"unordered {kind}, short")
            h.free()
            ds.free()
    finally:
        ctx.set_debug(0)
    d.free()


# ------------------------------------------------------------------ batched operators

def test_kmer_hash_batch(ctx, survey_vectors):
    for v in survey_vectors["kmer_hash"]:
        _, bits = orc.kmer_encode(v["kmer"])
        got = int(ctx.kmer_hash(np.array([bits], dtype=np.uint64))[0])
        assert got == v["hash"] & 0xFFFFFFFF
    rng = np.random.default_rng(3)
    keys = rng.integers(0, 1 << 63, size=100_000, dtype=np.uint64) * np.uint64(2) + np.uint64(1)
    want = np.array([orc.lib().orc_kmer_hash(int(x)) for x in keys[:5000]], dtype=np.uint32)
    got = ctx.kmer_hash(keys)
    assert_same(got[:5000], want, "kmer_hash")


def test_kmer_match_batch(ctx, pkg):
    n, k = 50_000, 21
    words = orc.synth_words(6, n)
    keys = orc.generate_kmers(words, n, k, faithful=False)
    for pat in PATTERNS_21[:3]:
        _, wp = orc.generate_kmers_contains(words, n, k, pat)
        want = np.zeros(len(keys), dtype=bool)
        want[wp.astype(np.int64)] = True
        assert np.array_equal(ctx.kmer_match(keys, k, pkg.Filter.contains(pat)), want)
    ln, bits = orc.kmer_encode("ACG")
    _, wp = orc.generate_kmers_starts_with(words, n, k, ln, bits)
    want = np.zeros(len(keys), dtype=bool)
    want[wp.astype(np.int64)] = True
    assert np.array_equal(ctx.kmer_match(keys, k, pkg.Filter.starts_with(ln, bits)), want)
    # kmer = q over the array (kmer_eq, dna.c:655-668): the rows that hold one given k-mer; a q of another length never equals
    q = int(keys[777])
    assert np.array_equal(ctx.kmer_match(keys, k, pkg.Filter.equals(k, q)), keys == np.uint64(q))
    assert ctx.kmer_match(keys, k, pkg.Filter.equals(k, q)).sum() >= 1
    assert not ctx.kmer_match(keys, k, pkg.Filter.equals(k - 1, q & ((1 << (2 * (k - 1))) - 1))).any()
    keys8 = orc.generate_kmers(words, n, 8, faithful=False)
    q8 = int(keys8[99])
    got8 = ctx.kmer_match(keys8, 8, pkg.Filter.equals(8, q8))
    assert np.array_equal(got8, keys8 == np.uint64(q8)) and got8.sum() >= 1
    with pytest.raises(pkg.DnaGpuError):
        ctx.kmer_match(keys, k, pkg.Filter.contains("ACG"))


# ------------------------------------------------------------------ multi-GPU step 1 (owner partition)

@pytest.mark.parametrize("n_owners", [1, 2, 3, 8])
def test_partition_by_owner(ctx, n_owners):
    import ctypes as C
    n, k = 400_000, 31
    d = ctx.synth(0xD2A0003, n)
    words = d.download()
    keys = orc.generate_kmers(words, n, k, faithful=False)
    for first, count in ((0, len(keys)), (1000, 250_000)):
        ptr, offs = ctx.partition_kmers(d, k, first, count, n_owners)
        assert int(offs[0]) == 0 and int(offs[-1]) == count and np.all(np.diff(offs.astype(np.int64)) >= 0)
        # read the device buffer back through a hist-free path: count each owner's slice separately
        bits = 10
        sub = keys[first:first + count]
        owner = ((sub >> np.uint64(2 * k - bits)).astype(np.uint64) * np.uint64(n_owners)) >> np.uint64(bits)
        for o in range(n_owners):
            lo, hi = int(offs[o]), int(offs[o + 1])
            want = np.sort(sub[owner == o])
            assert hi - lo == len(want), f"owner {o}: {hi - lo} keys, oracle {len(want)}"
            if hi > lo:
                import importlib
                # (shard_math, not sharded: importing torch here would put a second ROCm runtime into this process)
                sh = importlib.import_module(load_package().__name__ + ".shard_math")
                kmin, kmax = sh.owner_key_range(k, o, n_owners)
                assert int(want[0]) >= kmin and int(want[-1]) <= kmax
                if o % 2:     # both entry points
                    h = ctx.count_keys_device(C.c_void_p(ptr + lo * 8), hi - lo, k)
                else:
                    h = ctx.count_keys_device_in_range(C.c_void_p(ptr + lo * 8), hi - lo, k, kmin, kmax)
                ok, oc = orc.count_keys(want)
                check_hist(h, ok, oc, f"owner {o}/{n_owners} slice")
                h.free()
        ctx.buffer_free(ptr)
    d.free()


@pytest.mark.parametrize("n_owners", [2, 3, 8])
def test_count_kmers_owned(ctx, n_owners):
    # every owner scans the whole sequence and keeps its key range; the union is the full histogram
    for n, k, motif in ((500_000, 31, 0), (300_000, 21, 1000), (120_000, 8, 0), (40_000, 3, 0)):
        d = ctx.synth(77, n, motif_len=motif)
        words = d.download()
        keys = orc.generate_kmers(words, n, k, faithful=False)
        bits = min(2 * k, 10)
        owner = ((keys >> np.uint64(2 * k - bits)) * np.uint64(n_owners)) >> np.uint64(bits)
        all_k, all_c = [], []
        for o in range(n_owners):
            h = ctx.count_kmers_owned(d, k, o, n_owners)
            ok, oc = orc.count_keys(keys[owner == o])
            check_hist(h, ok, oc, f"owned n={n} k={k} owner {o}/{n_owners}")
            assert h.total == int((owner == o).sum())
            gk, gc = h.download()
            all_k.append(gk)
            all_c.append(gc)
            h.free()
        fk, fc = orc.count_keys(keys)
        assert_same(np.concatenate(all_k), fk, "owners concatenated = global keys")
        assert_same(np.concatenate(all_c), fc, "owners concatenated = global counts")
        d.free()


def pack_codes(codes):
    """2-bit codes (A=0 T=1 C=2 G=3) -> packed words in the reference's layout (dna.c:114-128)"""
    n = len(codes)
    pad = np.zeros((n + 31) // 32 * 32, dtype=np.uint64)
    pad[:n] = codes
    sh = np.arange(32, dtype=np.uint64) * np.uint64(2)
    return (pad.reshape(-1, 32) << sh).sum(axis=1, dtype=np.uint64)


@pytest.mark.parametrize("kind", ["scattered", "all-G", "many"])
@pytest.mark.parametrize("k", [32, 31, 21])
def test_count_unordered_copies_are_counted_in_place(ctx, pkg, kind, k):
    """sk_count_clean's buckets in which SOME records share k-mers: the sharing records' k-mers are counted in the record
    table's slots, the others leave as (key, 1).  scattered: 150-base pieces copied to random places (a few sharing records
    in nearly every bucket); all-G: copies of a piece around a run of 40 G -- at k = 32 the k-mer GG..G, the value of an
    empty slot, is counted beside the table; many: one piece copied 3000 times (more sharing k-mers than the slots take:
    those buckets go to sk_count)."""
    n = 5_000_000
    rng = np.random.default_rng(0xC0B1E5 + k)
    c = rng.integers(0, 4, n, dtype=np.uint8)
    if kind == "scattered":
        for _ in range(n // 7500):
            a, b = rng.integers(0, n - 150, 2)
            c[b:b + 150] = c[a:a + 150].copy()
    elif kind == "all-G":
        piece = rng.integers(0, 4, 140, dtype=np.uint8)
        piece[50:90] = 3
        for b in rng.integers(0, n - 140, 40):
            c[b:b + 140] = piece
    else:
        piece = rng.integers(0, 4, 100, dtype=np.uint8)
        for b in rng.integers(0, n - 100, 3000):
            c[b:b + 100] = piece
    words = pack_codes(c)
    d = ctx.upload(words, n)
    ok, oc = orc.count_kmers(words, n, k)
    if kind == "all-G" and k == 32:
        assert ok[-1] == np.uint64(0xFFFFFFFFFFFFFFFF) and oc[-1] >= 40
    ctx.set_debug(pkg.DEBUG_FORCE_SUPERKMER)
    try:
        ctx.set_profiling(True)
        h = ctx.count_kmers_unordered(d, k)
        names = [nm for nm, _ in ctx.last_phase_times()]
        ctx.set_profiling(False)
        assert "sk_count" in names, names
        check_hist_unordered(h, ok, oc, f"copies {kind} k={k}")
        h.free()
    finally:
        ctx.set_debug(0)
        ctx.set_profiling(False)
    d.free()


@pytest.mark.parametrize("k", [32, 26, 20])
def test_count_unordered_near_copies_every_key_width(ctx, pkg, k):
    """the expansion's keys travel mixed (key_mix is a bijection of the 2k-bit keys: 64 bits at k = 32, where the mixed
    all-ones key is counted beside the leaves' tables) -- a repeat family's near-copies, the engine forced so that its
    buckets are oversize at this size"""
    n = 4_000_000
    words = genome_like_words(n, 0x77 + k)
    d = ctx.upload(words, n)
    ok, oc = orc.count_kmers(words, n, k)
    ctx.set_debug(pkg.DEBUG_FORCE_SUPERKMER)
    try:
        h = ctx.count_kmers_unordered(d, k)
        check_hist_unordered(h, ok, oc, f"near copies k={k}")
        h.free()
    finally:
        ctx.set_debug(0)
    d.free()


@pytest.mark.parametrize("motif", [1000, 64, 1, 100000])
def test_count_unordered_repeats_at_size(ctx, motif):
    """repeat-rich inputs at 600 Mbase with the engine's own thresholds (heavy mid buckets by k-mers or records, sliced big
    buckets, merges): the unordered histogram's digest equals the ordered engine's"""
    n, k = 600_000_000, 31
    d = ctx.synth(0xD2A0007 + motif, n, motif_len=motif)
    ht = ctx.count_kmers(d, k)
    want = ht.summary()
    ht.free()
    hu = ctx.count_kmers_unordered(d, k)
    assert hu.summary() == want
    assert hu.total == n - k + 1
    hu.free()
    d.free()
    ctx.trim()


# ------------------------------------------------------------------ the unordered count in two halves (record exchange)

@pytest.mark.parametrize("world,n,k,motif", [(1, 3_000_000, 31, 0), (2, 3_000_017, 31, 0), (3, 2_000_003, 27, 0), (8, 5_000_000, 29, 0),
                                             (2, 100_000, 23, 0), (4, 1_000, 32, 0), (2, 4_000_000, 31, 1000), (3, 3_000_000, 31, 7),
                                             (2, 2_000_000, 25, 1), (3, 3_000_000, 21, 0), (2, 2_000_000, 22, 1000)])
def test_records_exchange_one_process(ctx, pkg, world, n, k, motif):
    """dnagpu_sk_records + dnagpu_count_records, the exchange done by hand in one process: every "rank" cuts the records
    of its own rows (its shard + the k-1 base halo, the global row count fixing the bucket geometry), every owner counts
    the pieces of its buckets from all ranks; the owners' groups are disjoint and together the oracle's histogram.
    motif > 0: repeat-rich input (heavy buckets; more than half the k-mers heavy: everything is expanded to keys)."""
    import importlib
    sh = importlib.import_module(pkg.__name__ + ".shard_math")
    seed = 0xD2A0003 + n
    words = orc.synth_words_repeat(seed, n, motif) if motif else orc.synth_words(seed, n)
    ok, oc = orc.count_kmers(words, n, k)
    rows = n - k + 1
    n_buckets = ctx.sk_buckets(rows, k)
    assert n_buckets >= 1
    recs = []
    for first, cnt, lo, hi in sh.shard_ranges(n, k, world):
        w = np.ascontiguousarray(words[lo // 32:(hi + 31) // 32]) if hi > lo else np.zeros(0, dtype=np.uint64)
        d = ctx.upload(w, hi - lo)
        r = ctx.sk_records(d, k, 0, cnt, rows)
        assert r.n_buckets == n_buckets and int(r.offsets[0]) == 0 and np.all(np.diff(r.offsets.astype(np.int64)) >= 0)
        d.free()
        recs.append(r)
    gk, gc, total = [], [], 0
    for lo_b, hi_b in sh.bucket_owner_ranges(n_buckets, world):
        pieces = [(r.device_ptr + 16 * int(r.offsets[b]), int(r.offsets[b + 1] - r.offsets[b]), b)
                  for r in recs for b in range(lo_b, hi_b) if r.n_records]
        h = ctx.count_records(pieces, k, rows)
        assert not h.is_sorted or h.distinct == 0
        a, c = h.download()
        gk.append(a)
        gc.append(c)
        total += h.total
        h.free()
    for r in recs:
        r.free()
    gk, gc = np.concatenate(gk), np.concatenate(gc)
    order = np.argsort(gk, kind="stable")
    assert total == rows
    assert_same(gk[order], ok, f"records exchange world={world} keys")
    assert_same(gc[order], oc, f"records exchange world={world} counts")
    with pytest.raises(pkg.DnaGpuError):
        ctx.count_records([(0, 5, n_buckets)], k, rows)          # a bucket the geometry does not have


def test_count_records_mostly_repeats_expand_everything(ctx, pkg):
    """dnagpu_count_records over records that are almost all heavy (a periodic sequence), with the heavy mid buckets
    expanded as a whole (DNAGPU_DEBUG_HEAVY_EXPAND): levels 1-2 report the set skewed, every coarse bucket becomes one
    key node (no final bucket at all) and the ordinary levels count the keys."""
    k = 27
    text = ("ACGTTGCAATCCGA" * 25_000) + "".join("ACGT"[(i * 7 + i // 3) % 4] for i in range(50_001))
    words, n = orc.dna_encode(text)
    ok, oc = orc.count_kmers(words, n, k)
    rows = n - k + 1
    d = ctx.upload(words, n)
    ctx.set_debug(pkg.DEBUG_FORCE_SUPERKMER | pkg.DEBUG_HEAVY_EXPAND)
    try:
        r = ctx.sk_records(d, k, 0, rows, rows)
        pieces = [(r.device_ptr + 16 * int(r.offsets[b]), int(r.offsets[b + 1] - r.offsets[b]), b) for b in range(r.n_buckets)]
        h = ctx.count_records(pieces, k, rows)
        hu = ctx.count_kmers_unordered(d, k)          # (the same input through the single-GPU entry point: the tree from scratch)
    finally:
        ctx.set_debug(0)
    assert h.total == rows
    check_hist_unordered(h, ok, oc, "records of a mostly periodic sequence, everything expanded")
    check_hist_unordered(hu, ok, oc, "mostly periodic sequence, unordered entry point with HEAVY_EXPAND")
    for o in (h, hu, r, d):
        o.free()


def test_count_records_fingerprint_collisions(ctx, pkg):
    """sk_count's table slots hold a 19-bit fingerprint, not the key: keys that share slot AND fingerprint must still be
    told apart (the claimant's key is re-derived from the staged records and compared in full).  Hand-made records of
    one k-mer each, all in one final bucket: key ^ (m | m << 32) for m < 128 leaves lo ^ hi, and with it both the slot
    and the fingerprint (functions of lo ^ hi), unchanged -- 128 distinct keys on one probe chain with one fingerprint --
    some of them several times, mixed with random keys."""
    k, rows = 31, 50_000_000                       # (rows only fixes the bucket geometry)
    nb = ctx.sk_buckets(rows, k)
    rng = np.random.default_rng(7)
    kmask = (1 << (2 * k)) - 1
    base = int(rng.integers(0, 1 << 62)) & kmask & ~((127 << 32) | 127)
    fam = [base ^ (m | (m << 32)) for m in range(128)]
    keys = list(fam)                               # at most 512 records, so that the bucket is sk_count's (not the tree's)
    for i, kv in enumerate(fam[:60]):
        keys += [kv] * (1 + i % 3)                 # copies
    keys += [int(x) & kmask for x in rng.integers(0, 1 << 62, 200)]
    keys = np.array(keys, dtype=np.uint64)
    assert len(keys) <= 512
    rng.shuffle(keys)
    d1, d2, bucket = 5, 3, nb - 1
    recs = np.empty(2 * len(keys), dtype=np.uint64)
    recs[0::2] = keys                              # lo: the k-mer's 31 bases; hi: no more bases, len - 1 = 0
    recs[1::2] = np.uint64((d1 << 49) | (d2 << 59))
    buf = ctx.buffer_alloc(recs.nbytes)
    ctx.upload_u64(buf, recs)
    h = ctx.count_records([(buf, len(keys), bucket)], k, rows)
    ok, oc = np.unique(keys, return_counts=True)
    assert h.total == len(keys)
    check_hist_unordered(h, ok, oc.astype(np.uint64), "crafted fingerprint family")
    assert h.extent == len(keys)                   # sk_count's placement: one slot per k-mer, copies leave padding
    h.free()
    # the same keys as 3 x more records than sk_count takes: the expansion + hashed-leaves path
    big = np.concatenate([keys, keys, keys, keys])
    recs = np.empty(2 * len(big), dtype=np.uint64)
    recs[0::2] = big
    recs[1::2] = np.uint64((d1 << 49) | (d2 << 59))
    buf2 = ctx.buffer_alloc(recs.nbytes)
    ctx.upload_u64(buf2, recs)
    h = ctx.count_records([(buf2, len(big), bucket)], k, rows)
    check_hist_unordered(h, ok, (4 * oc).astype(np.uint64), "crafted fingerprint family, oversize bucket")
    h.free()
    ctx.buffer_free(buf2)
    ctx.buffer_free(buf)


@pytest.mark.parametrize("kind", ["many-distinct", "sliced-few", "sliced-overflow", "sliced-polyG"])
def test_count_records_big_buckets(ctx, pkg, kind):
    """sk_count_big on hand-made buckets (records of one k-mer each, one final bucket): more distinct keys than its table
    holds (the bucket must come back through the expansion), a bucket of several slices with few distinct keys (partial
    areas + merge), a sliced bucket that overflows in the merge, and the all-ones key (k = 32 poly-G), which the table
    cannot hold as a key."""
    k = 32 if kind == "sliced-polyG" else 31
    rows = 50_000_000
    nb = ctx.sk_buckets(rows, k)
    rng = np.random.default_rng(11)
    kmask = (1 << (2 * k)) - 1
    if kind == "many-distinct":
        keys = rng.integers(0, 1 << 62, 9000, dtype=np.uint64) & np.uint64(kmask)
    elif kind == "sliced-few":
        pool = rng.integers(0, 1 << 62, 37, dtype=np.uint64) & np.uint64(kmask)
        keys = pool[rng.integers(0, len(pool), 150_000)]
    elif kind == "sliced-overflow":
        pool = rng.integers(0, 1 << 62, 7000, dtype=np.uint64) & np.uint64(kmask)
        keys = pool[rng.integers(0, len(pool), 150_000)]
    else:
        pool = np.concatenate([np.array([kmask], dtype=np.uint64), rng.integers(0, 1 << 63, 20, dtype=np.uint64) & np.uint64(kmask)])
        keys = pool[rng.integers(0, len(pool), 140_000)]
    d1, d2, bucket = 9, 1, 0
    recs = np.empty(2 * len(keys), dtype=np.uint64)
    if k == 31:
        recs[0::2] = keys
        recs[1::2] = np.uint64((d1 << 49) | (d2 << 59))
    else:                                          # 32 bases: all of lo; hi carries no base
        recs[0::2] = keys
        recs[1::2] = np.uint64((d1 << 49) | (d2 << 59))
    buf = ctx.buffer_alloc(recs.nbytes)
    ctx.upload_u64(buf, recs)
    h = ctx.count_records([(buf, len(keys), bucket)], k, rows)
    ok, oc = np.unique(keys, return_counts=True)
    assert h.total == len(keys)
    check_hist_unordered(h, ok, oc.astype(np.uint64), f"crafted big bucket: {kind}")
    h.free()
    ctx.buffer_free(buf)


def test_pool_guard_bands_catch_a_write_past_the_end(pkg):
    """DNAGPU_DEBUG_GUARD_POOL: a write past the end of a work buffer -- here made on purpose through
    dnagpu_buffer_upload -- fails the next dnagpu_synchronize, naming the buffer's size; a clean run does not."""
    with pkg.Context(0) as c:
        c.set_debug(pkg.DEBUG_GUARD_POOL)
        d = c.synth(3, 300_000)
        h = c.count_kmers_unordered(d, 31)
        c.synchronize()                                   # every band of the count's work buffers is intact
        h.free()
        buf = c.buffer_alloc(1000)                        # 125 words
        c.upload_u64(buf, np.arange(125, dtype=np.uint64))
        c.synchronize()
        c.upload_u64(buf, np.arange(126, dtype=np.uint64))   # one word too many
        with pytest.raises(pkg.DnaGpuError) as ei:
            c.synchronize()
        assert "past the end of a 1000-byte" in str(ei.value)
        c.set_debug(0)
        c.buffer_free(buf)
        d.free()


# ------------------------------------------------------------------ GROUP BY over a table of sequences (test.sql:140-150)

def table_of_sequences(seed, n_seqs, lo, hi):
    """n_seqs sequences of lo .. hi bases back to back in one packed stream (+ a few empty ones and some shorter than any k):
    -> (words, starts)"""
    rng = np.random.default_rng(seed)
    lens = rng.integers(lo, hi + 1, n_seqs)
    if n_seqs >= 10:
        lens[rng.integers(0, n_seqs, max(1, n_seqs // 50))] = 0          # empty rows
        lens[rng.integers(0, n_seqs, max(1, n_seqs // 50))] = rng.integers(1, 8)
    starts = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
    n = int(starts[-1])
    return orc.synth_words(seed, n), starts


@pytest.mark.parametrize("n_seqs,lo,hi", [(1, 5000, 5000), (2, 50, 10_000), (1000, 50, 10_000), (1_000_000, 50, 250),
                                         (200_000, 140, 160), (3, 1, 40), (40_000, 20, 64)])
def test_count_kmers_batch_table_of_sequences(ctx, pkg, n_seqs, lo, hi):
    """dnagpu_count_kmers_batch: GROUP BY kmer over FROM table, LATERAL generate_kmers(sequence, k) -- every sequence's own
    rows, none across two sequences, none for a sequence shorter than k -- against the oracle's per-sequence
    generate_kmers + hash aggregate.  Long k-mers through the super-k-mer engine (forced for the short tables), short
    ones and the default choice through the compacted keys; k = 10 is the reference's own (test.sql:143)."""
    words, starts = table_of_sequences(0xBA7C4 + n_seqs, n_seqs, lo, hi)
    n = int(starts[-1])
    d = ctx.upload(words, n)
    want = {}
    # (the long tables with two k only: the oracle's hash aggregate of 1.5 x 10^8 rows takes most of a minute per k)
    for k in ((31, 21, 32, 25, 20, 10, 3) if n_seqs < 200_000 else (31, 10)):
        ok, oc = want[k] = orc.count_keys(orc.generate_kmers_table(words, starts, k))
        for forced in (True, False):
            if forced and k < 20:
                continue
            ctx.set_debug(pkg.DEBUG_FORCE_SUPERKMER | pkg.DEBUG_SLAB0 if forced else 0)
            try:
                h = ctx.count_kmers_batch(d, starts, k)
            finally:
                ctx.set_debug(0)
            what = f"table of {n_seqs} sequences ({lo}..{hi} bases), k={k}, {'records' if forced else 'default engine'}"
            assert h.total == int(oc.sum()), what
            if h.is_sorted:
                check_hist(h, ok, oc, what)
            else:
                check_hist_unordered(h, ok, oc, what)
            h.free()
    # the same table made resident (dnagpu_dna_set_sequences) and counted for several k (dnagpu_count_kmers_table)
    d.set_sequences(starts)
    assert d.n_sequences == n_seqs
    for k in (31, 20, 10):
        if k not in want:
            continue
        ok, oc = want[k]
        for forced in (True, False):
            if forced and k < 20:
                continue
            ctx.set_debug(pkg.DEBUG_FORCE_SUPERKMER if forced else 0)
            try:
                h = ctx.count_kmers_table(d, k)
            finally:
                ctx.set_debug(0)
            what = f"resident table of {n_seqs} sequences, k={k}, {'records' if forced else 'default engine'}"
            assert h.total == int(oc.sum()), what
            check_hist(h, ok, oc, what) if h.is_sorted else check_hist_unordered(h, ok, oc, what)
            h.free()
    d.free()


@pytest.mark.parametrize("k", [31, 10, 32])
def test_hist_merge_adds_the_batches_of_a_table(ctx, pkg, k):
    """dnagpu_hist_merge: a table counted in two batches (dnagpu_count_kmers_batch each) and a repeat-rich single sequence
    (an unordered histogram with count-0 padding), merged == the oracle's count over all rows; merging with an empty
    histogram changes nothing; the all-ones key (32 G's) survives the merge."""
    words, starts = table_of_sequences(0x3E46E + k, 30_000, 60, 400)
    n = int(starts[-1])
    cut = 17_000                                       # sequences [0, cut) and [cut, ...) as two packed streams
    b_cut = int(starts[cut])
    seqs = orc.dna_decode(words, n)
    wa, na = orc.dna_encode(seqs[:b_cut])
    wb, nb = orc.dna_encode(seqs[b_cut:])
    da, db = ctx.upload(wa, na), ctx.upload(wb, nb)
    ctx.set_debug(pkg.DEBUG_FORCE_SUPERKMER if k >= 21 else 0)
    try:
        ha = ctx.count_kmers_batch(da, starts[:cut + 1], k)
        hb = ctx.count_kmers_batch(db, starts[cut:] - np.uint64(b_cut), k)
        # a third histogram: one sequence with a tiled motif (copies: padding slots in the unordered arrays) and 32 G's
        wr = orc.synth_words_repeat(5, 400_000, 1000).copy()
        wr[100:104] = np.uint64(0xFFFFFFFFFFFFFFFF)
        dr = ctx.upload(wr, 400_000)
        hr = ctx.count_kmers_unordered(dr, k)
    finally:
        ctx.set_debug(0)
    keys = np.concatenate([orc.generate_kmers_table(words, starts, k), orc.generate_kmers(wr, 400_000, k, faithful=False)])
    ok, oc = orc.count_keys(keys)
    hab = ha.merge(hb)
    hall = hab.merge(hr)
    assert hall.total == len(keys) and not hall.is_sorted
    check_hist_unordered(hall, ok, oc, f"merged histograms, k={k}")
    if k == 32:
        assert np.uint64(0xFFFFFFFFFFFFFFFF) in ok     # (the key the merge table cannot hold goes through its own counter)
    empty = ctx.count_kmers_batch(da, starts[:cut + 1], k).merge(ha)      # (sanity: a + a doubles every count)
    ok2, oc2 = orc.count_keys(orc.generate_kmers_table(wa, starts[:cut + 1], k))
    check_hist_unordered(empty, ok2, oc2 * np.uint64(2), f"a + a, k={k}")
    for h in (ha, hb, hr, hab, hall, empty):
        h.free()
    for d in (da, db, dr):
        d.free()


def test_count_kmers_batch_is_the_plain_count_for_one_sequence_and_checks_its_arguments(ctx, pkg):
    n = 300_000
    words = orc.synth_words(77, n)
    d = ctx.upload(words, n)
    ok, oc = orc.count_kmers(words, n, 31)
    h = ctx.count_kmers_batch(d, [0, n], 31)
    check_hist(h, ok, oc, "one sequence") if h.is_sorted else check_hist_unordered(h, ok, oc, "one sequence")
    h.free()
    for bad in ([0, n - 1], [1, n], [0, 200, 100, n], [0]):
        with pytest.raises(pkg.DnaGpuError) as ei:
            ctx.count_kmers_batch(d, bad, 31)
        assert ei.value.code == 5                  # DNAGPU_ERR_BAD_ARG
    with pytest.raises(pkg.DnaGpuError) as ei:
        ctx.count_kmers_batch(d, [0, n], 33)
    assert ei.value.code == 1 and "between 1 and 32" in str(ei.value)      # dna.c:773
    with pytest.raises(pkg.DnaGpuError) as ei:     # no sequences set on this stream
        ctx.count_kmers_table(d, 31)
    assert ei.value.code == 5
    for bad in ([0, n - 1], [0, 200, 100, n]):
        with pytest.raises(pkg.DnaGpuError) as ei:
            d.set_sequences(bad)
        assert ei.value.code == 5 and d.n_sequences == 0
    d.set_sequences([0, 100, n])
    d.set_sequences([0, n])                        # (replaces the first set: one sequence, the plain count)
    h = ctx.count_kmers_table(d, 31)
    check_hist(h, ok, oc, "one sequence, resident") if h.is_sorted else check_hist_unordered(h, ok, oc, "one sequence, resident")
    h.free()
    with pytest.raises(pkg.DnaGpuError) as ei:
        ctx.count_kmers_table(d, 0)
    assert ei.value.code == 1
    h = ctx.count_kmers_batch(d, [0, 10, 20, 30, n - 5, n], 32)
    keys = orc.generate_kmers_table(words, np.array([0, 10, 20, 30, n - 5, n], dtype=np.uint64), 32)
    ok, oc = orc.count_keys(keys)
    check_hist(h, ok, oc, "short sequences around a long one") if h.is_sorted else check_hist_unordered(h, ok, oc, "short sequences around a long one")
    h.free()
    d.free()


# ------------------------------------------------------------------ multi-GPU count through the C-ABI (one process)

def rank_device_maps(pkg, n_ranks):
    """device lists for n_ranks ranks: all on device 0 (what a one-GPU test box can run), and -- on a box with two or
    more devices -- the ranks spread over min(n_ranks, device_count) DISTINCT devices, so that the peer copies (and the
    RCCL communicator) really cross a link."""
    maps = [([0] * n_ranks, "shared")]
    n_dev = pkg.device_count()
    if n_dev >= 2 and n_ranks >= 2:
        use = min(n_ranks, n_dev)
        maps.append(([r % use for r in range(n_ranks)], f"{use} devices"))
    return maps


@pytest.mark.parametrize("n_ranks", [1, 2, 3, 8])
def test_count_multi_one_process(pkg, n_ranks):
    """dnagpu_count_multi: N ranks driven from one process (every rank mapped to device 0 here, copy transport;
    chunk residency, the gather of the packed chunks and the per-rank owner counts are the product code).  The
    ranks' ascending downloads concatenated in rank order == the oracle's sorted histogram."""
    n, seed = 3_000_017, 0xD2A0003
    words = orc.synth_words(seed, n)
    for devices, where in rank_device_maps(pkg, n_ranks):
        distinct = len(set(devices)) == n_ranks and n_ranks > 1
        # distinct devices: whatever DNAGPU_MULTI_AUTO gives (RCCL all-gather / reduce when the communicator can be made)
        with pkg.Multi(devices, pkg.MULTI_AUTO if distinct else pkg.MULTI_COPY) as m:
            assert m.transport in ("copy", "rccl") and (distinct or m.transport == "copy")
            for make in ("synth", "upload"):
                d = m.synth(seed, n) if make == "synth" else m.upload(words, n)
                for k, first, count in ((31, 0, None), (21, 1000, 2_000_000), (8, 0, None), (3, 5, 1_000_000)):
                    ok, oc = orc.count_keys(orc.generate_kmers(words, n, k, faithful=False)[first:None if count is None else first + count])
                    hs = m.count(d, k, first, count)
                    gk = np.concatenate([h.download()[0] for h in hs])
                    gc = np.concatenate([h.download()[1] for h in hs])
                    assert sum(h.total for h in hs) == int(oc.sum())
                    assert_same(gk, ok, f"multi {n_ranks} ranks on {where} ({make}) k={k} keys")
                    assert_same(gc, oc, f"multi {n_ranks} ranks on {where} ({make}) k={k} counts")
                    for h in hs:
                        h.free()
                m.dna_free(d)


@pytest.mark.parametrize("n_ranks,parts", [(1, 1), (2, 2), (3, 3), (8, 2), (8, 1), (2, 8)])
def test_count_multi_unordered_one_process(pkg, n_ranks, parts):
    """dnagpu_count_multi_unordered: the record exchange from one process (every rank on device 0: chunk residency, the
    halo word, per-rank records, the owners' pulls of their buckets' pieces on the transfer stream, the counting of a
    bucket group behind its event are the product code).  The ranks' groups are disjoint and together the oracle's
    histogram, whatever the number of bucket groups per owner; short k-mers take the ordered paths."""
    n, seed = 3_000_017, 0xD2A0003
    words = orc.synth_words(seed, n)
    runs = [(devices, where, 0) for devices, where in rank_device_maps(pkg, n_ranks)]
    if len(runs) > 1 and len(set(runs[1][0])) == n_ranks:
        runs.append((runs[1][0], runs[1][1] + ", RCCL send/recv", 1))      # the same exchange through ncclSend / ncclRecv
    for devices, where, via_rccl in runs:
        _count_multi_unordered_case(pkg, n_ranks, parts, devices, where, via_rccl, n, seed, words)


def _count_multi_unordered_case(pkg, n_ranks, parts, devices, where, via_rccl, n, seed, words):
    shared = len(set(devices)) < len(devices)
    with pkg.Multi(devices, pkg.MULTI_COPY if shared else pkg.MULTI_AUTO) as m:
        m.set_parts(parts)
        for c in m.ranks:
            c.set_profiling(True)           # (rank_phase_times below)
        if via_rccl:
            if m.transport != "rccl":
                return                      # (no communicator on this box: the copy transport was the first run)
            m.set_exchange_rccl(1)
        if parts == 2 and n_ranks == 2 and shared:
            m.emulate_link(50.0)            # the rehearsal delay kernel on the transfer stream: timing only
        for make in ("synth", "upload"):
            d = m.synth(seed, n) if make == "synth" else m.upload(words, n)
            for k, first, count in ((31, 0, None), (27, 1000, 2_000_000), (23, 31, None), (32, 0, 1_234_567), (21, 5, None), (8, 0, None)):
                ok, oc = orc.count_keys(orc.generate_kmers(words, n, k, faithful=False)[first:None if count is None else first + count])
                hs = m.count_unordered(d, k, first, count)
                gk = np.concatenate([h.download()[0] for h in hs])
                gc = np.concatenate([h.download()[1] for h in hs])
                assert sum(h.total for h in hs) == int(oc.sum())
                order = np.argsort(gk, kind="stable")
                assert_same(gk[order], ok, f"multi unordered {n_ranks} ranks x {parts} parts ({make}) k={k} keys")
                assert_same(gc[order], oc, f"multi unordered {n_ranks} ranks x {parts} parts ({make}) k={k} counts")
                # the digest over all parts, and windows of the read order that straddle part boundaries
                t = [0, 0, 0, 0]
                for h in hs:
                    assert 1 <= h.n_parts <= parts
                    t = [(a + b) & ((1 << 64) - 1) for a, b in zip(t, h.summary())]
                assert tuple(t) == orc.hist_summary(ok, oc)
                h0 = max(hs, key=lambda h: h.n_parts)
                if h0.distinct > 10:
                    fk, fc = h0.download()
                    a, b = h0.distinct // 3, h0.distinct - h0.distinct // 4
                    wk, wc = h0.download(a, b - a)
                    assert_same(wk, fk[a:b], "window of a histogram of several parts: keys")
                    assert_same(wc, fc[a:b], "window of a histogram of several parts: counts")
                if k >= 21:
                    lt = m.last_times()
                    assert lt["parts"] == parts and lt["total_ms"] > 0
                    assert m.exchange_transport == ("rccl-sendrecv" if via_rccl else "peer-copy"), where
                    if count is None:           # (a window may lie in one rank's chunk: then nothing travels)
                        assert (lt["bytes_moved"] > 0) == (n_ranks > 1)
                    # a rank's phases = its record pass (level 0), then its owner phase (some owner counts with sk_count)
                    names = [[nm for nm, _ in m.rank_phase_times(r)] for r in range(n_ranks)]
                    assert any(nm.startswith("sk_scatter0") for nm in names[0]), names[0]
                    assert any("sk_count" in nms for nms in names), names
                for h in hs:
                    h.free()
            m.dna_free(d)


def test_bench_gpus2_one_process_rehearsal():
    """`python bench.py --gpus 2` started the way the driver starts it (a plain command, no torchrun): on a one-GPU box
    the two ranks share device 0 (copy transport, "rehearsal": true); the line keeps the contract's fields."""
    import json
    import subprocess
    import sys
    from __graft_entry__ import ROOT
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                          "--n-bases", "200000000"], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    line = json.loads(out.stdout.strip().splitlines()[-1])
    pkg = load_package()
    assert line["ranks"] == 2 and line["launcher"] == "one-process" and line["unit"] == "k-mers/s"
    assert line["n_gpus"] == min(2, pkg.device_count())
    assert line["config"]["distinct"] > 0 and line["value"] > 0 and line["scaling"] == "strong"
    assert line["roofline"] and line["exchange"]["parts"] >= 1
    # the line verifies itself: the ranks' digests summed; 200 Mbase is not a BASELINE config, so only sum(count) == rows
    assert line["digest_total_ok"] is True and line["digest_ok"] is None and line["digest"]["total"] == 200000000 - 31 + 1
    assert any(nm.startswith("sk_scatter0") for nm in line["phases_ms"]) and "sk_count" in line["phases_ms"]
    if pkg.device_count() < 2:
        assert line["rehearsal"] is True and line["exchange_transport"] == "peer-copy" and line["rccl_ranks"] == 0
        assert line["devices"] == [0, 0]


def test_bench_digest_check_flips_on_a_wrong_digest(tmp_path):
    """bench.py's self-check: config 3 (k = 31, 248,956,422 bases) prints digest_ok true against the oracle's digest and
    exits 0; the same run against a copy of the golden file with one value changed prints digest_ok false and exits 3."""
    import json
    import shutil
    import subprocess
    import sys
    from __graft_entry__ import ROOT
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--config", "3", "--steps", "1", "--warmup", "1", "--no-cpu-baseline"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    line = json.loads(out.stdout.strip().splitlines()[-1])
    assert line["digest_ok"] is True and line["digest_total_ok"] is True
    want = json.load(open(os.path.join(ROOT, "tests", "golden", "config_digests.json")))["3"]
    assert line["digest"] == {f: want[f] for f in ("total", "distinct", "unique", "checksum")}
    # a wrong expectation (the checker must be able to fail): BENCH_DIGESTS points bench.py at a doctored copy
    bad = json.load(open(os.path.join(ROOT, "tests", "golden", "config_digests.json")))
    bad["3"]["checksum"] ^= 1
    p = tmp_path / "digests.json"
    p.write_text(json.dumps(bad))
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=dict(os.environ, BENCH_DIGESTS=str(p)))
    assert out.returncode == 3
    line = json.loads(out.stdout.strip().splitlines()[-1])
    assert line["digest_ok"] is False and line["digest_expected"]["checksum"] == bad["3"]["checksum"]


def test_count_multi_rccl_one_rank(pkg, ctx):
    """the RCCL transport (librccl.so loaded on demand, ncclCommInitAll) with the one device a test box has"""
    ctx.trim()          # the module's context has pooled the full-size configs' work buffers: RCCL allocates on its own
    n, k, seed = 1_000_003, 31, 0xD2A0002
    words = orc.synth_words(seed, n)
    with pkg.Multi([0], pkg.MULTI_RCCL) as m:
        assert m.transport == "rccl"
        d = m.synth(seed, n)
        hs = m.count(d, k)
        ok, oc = orc.count_kmers(words, n, k)
        check_hist(hs[0], ok, oc, "multi rccl one rank")
        hs[0].free()
        hs = m.count(d, 8)                          # short k-mers: the table path (ncclReduce of the 4^k counters)
        ok, oc = orc.count_kmers(words, n, 8)
        check_hist(hs[0], ok, oc, "multi rccl one rank, dense")
        hs[0].free()
        m.dna_free(d)
    with pytest.raises(pkg.DnaGpuError):
        pkg.Multi([0, 0], pkg.MULTI_RCCL)          # RCCL needs distinct devices


def test_count_multi_unordered_rccl_sendrecv_one_rank(pkg, ctx):
    """The record exchange through RCCL (DNAGPU_MULTI_OPT_EXCHANGE_RCCL) with the one device a test box has: mode 2 sends
    a rank's OWN pieces through ncclSend / ncclRecv too, so one rank alone runs the group calls, the piece offsets and the
    landing layout; the groups are the oracle's.  Without a communicator the option is refused."""
    ctx.trim()
    n, seed = 3_000_017, 0xD2A0003
    words = orc.synth_words(seed, n)
    with pkg.Multi([0], pkg.MULTI_COPY) as m:
        with pytest.raises(pkg.DnaGpuError):
            m.set_exchange_rccl(1)
    with pkg.Multi([0], pkg.MULTI_RCCL) as m:
        m.set_exchange_rccl(2)
        d = m.synth(seed, n)
        for parts in (1, 3):
            m.set_parts(parts)
            for k in (31, 21):
                hs = m.count_unordered(d, k)
                assert m.exchange_transport == "rccl-sendrecv" and m.rccl_ranks == 1
                ok, oc = orc.count_kmers(words, n, k)
                check_hist_unordered(hs[0], ok, oc, f"rccl send/recv, one rank, {parts} parts, k={k}")
                hs[0].free()
        m.set_exchange_rccl(0)
        hs = m.count_unordered(d, 31)
        assert m.exchange_transport == "peer-copy"
        hs[0].free()
        m.dna_free(d)


# ------------------------------------------------------------------ BASELINE.json sizes (properties)

def oracle_digest(name):
    """(total, distinct, unique, checksum) of a BASELINE.json config at its FULL size as the CPU oracle counted it in the
    build container (tools/make_digests.py -> tests/golden/config_digests.json)."""
    import json
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "config_digests.json")) as f:
        e = json.load(f)[name]
    return e["total"], e["distinct"], e["unique"], e["checksum"]


def test_config2_k21_100M_against_oracle_summary(ctx):
    """configs[1]: k=21 over 100 Mbase synthetic, seed 0xD2A0001: full histogram digest vs oracle."""
    n, k, seed = 100_000_000, 21, 0xD2A0001
    d = ctx.synth(seed, n)
    h = ctx.count_kmers(d, k)
    total, distinct, unique, checksum = h.summary()
    assert total == n - k + 1
    assert (total, distinct, unique, checksum) == oracle_digest("2")
    hu = ctx.count_kmers_unordered(d, k)          # what bench.py --config 2 runs
    assert hu.summary() == oracle_digest("2")
    hu.free()
    words = orc.synth_words(seed, n)
    ok, oc = orc.count_kmers(words, n, k)
    assert (total, distinct, unique, checksum) == orc.hist_summary(ok, oc)
    gk, gc = h.download()
    assert_same(gk, ok, "config 2 keys")
    assert_same(gc, oc, "config 2 counts")
    h.free()
    d.free()


def test_config5_k21_100M_contains(ctx, pkg):
    """BASELINE config 5 at its stated size: qkmer @> fused into the k=21 extraction over the 100 Mbase
    config-2 sequence (contains, dna.c:1091-1135; row order test.sql:86-92): keys, positions and the
    total equal the oracle's, for a pattern of selectivity 1/4 in the middle of the k-mer, one at its
    start, and a selective one (1/256)."""
    n, k, seed = 100_000_000, 21, 0xD2A0001
    words = orc.synth_words(seed, n)
    d = ctx.synth(seed, n)
    for pat in ("NNNNNNNNNNWSNNNNNNNNN", "RYNNNNNNNNNNNNNNNNNNN", "ACNNNNNNNNNNNNNNNNGTN"):
        wk, wp = orc.generate_kmers_contains(words, n, k, pat)
        gk, gp, tot = ctx.generate_kmers_filtered(d, k, pkg.Filter.contains(pat))
        assert tot == len(wk), f"{pat}: {tot} matches, oracle {len(wk)}"
        assert_same(gk, wk, f"config 5 {pat} keys")
        assert_same(gp, wp, f"config 5 {pat} positions")
        del wk, wp, gk, gp
    d.free()


def test_config3_k31_chr1_scale_properties(ctx):
    """configs[2]: k=31 over 248,956,422 bases: sortedness, totals, and window additivity."""
    n, k, seed = 248_956_422, 31, 0xD2A0002
    d = ctx.synth(seed, n)
    h = ctx.count_kmers(d, k)
    total, distinct, unique, checksum = h.summary()
    assert total == n - k + 1
    # total / distinct / unique (test.sql:107-119) and the checksum over all groups: the ORACLE's, counted at this size
    assert (total, distinct, unique, checksum) == oracle_digest("3"), "ordered engine vs the oracle's digest"
    hu = ctx.count_kmers_unordered(d, k)          # what bench.py --config 3 runs (super-k-mer engine at this size)
    assert not hu.is_sorted
    assert hu.summary() == oracle_digest("3"), "unordered engine vs the oracle's digest"
    hu.free()
    gk, gc = h.download()
    assert np.all(gk[1:] > gk[:-1]), "keys not strictly ascending"
    assert int(gc.sum()) == total and int((gc == 1).sum()) == unique
    # uniform 62-bit keys: almost all distinct
    assert distinct > total - 100
    # spot check: 200 random positions' k-mers are present with count >= 1
    rng = np.random.default_rng(1)
    pos = rng.integers(0, n - k + 1, size=200)
    words = d.download()
    for p in pos:
        key = orc.generate_kmers(words, n, k, first=int(p), count=1, faithful=False)[0]
        i = int(np.searchsorted(gk, key))
        assert i < len(gk) and gk[i] == key
    h.free()
    d.free()


def test_config4_k31_3G_properties(ctx, pkg):
    """configs[3] at its full size on one GPU: k=31 over 3 Gbase, seed 0xD2A0003 (the bench workload).
    Size-independent properties: sum(count) = number of rows; groups strictly ascending inside and
    across download windows; the eight owners' histograms (the sharded path) are disjoint and their
    digests add up to the single-GPU digest (a checksum of checksums); rows owned add up to all rows."""
    n, k, seed = 3_000_000_000, 31, 0xD2A0003
    d = ctx.synth(seed, n)
    h = ctx.count_kmers(d, k)
    total, distinct, unique, checksum = h.summary()
    assert total == n - k + 1
    assert total - 100 < distinct <= total and unique <= distinct
    # the ORACLE's digest of the same 3 Gbase (30 key-space slices on the CPU, tools/make_digests.py): both engines and
    # both multi-GPU exchanges below are compared with it, not with each other
    assert (total, distinct, unique, checksum) == oracle_digest("4"), "ordered engine vs the oracle's digest"
    prev_last = None
    for first in (0, distinct // 3, distinct - 4_000_000):
        gk, gc = h.download(first, 4_000_000)
        assert np.all(gk[1:] > gk[:-1]), f"window at {first}: keys not strictly ascending"
        assert gc.min() >= 1
        if prev_last is not None:
            assert gk[0] > prev_last
        prev_last = int(gk[-1])
    h.free()
    M = (1 << 64) - 1
    t_sum = d_sum = u_sum = c_sum = 0
    last_key = -1
    for owner in range(8):
        ho = ctx.count_kmers_owned(d, k, owner, 8)
        t, dd, u, c = ho.summary()
        assert ho.total == t
        fk, _ = ho.download(0, 1)
        lk, _ = ho.download(dd - 1, 1)
        assert int(fk[0]) > last_key, "owners' key ranges overlap or are out of order"
        last_key = int(lk[0])
        t_sum, d_sum, u_sum, c_sum = t_sum + t, d_sum + dd, u_sum + u, (c_sum + c) & M
        ho.free()
    assert (t_sum, d_sum, u_sum, c_sum) == (total, distinct, unique, checksum)
    # the unordered entry point (super-k-mer engine: the bench's default workload): the same groups, i.e. the same digest
    hu = ctx.count_kmers_unordered(d, k)
    assert not hu.is_sorted, "3 Gbase k=31 is expected to go through the super-k-mer engine"
    assert hu.summary() == oracle_digest("4"), "unordered engine vs the oracle's digest"
    hu.free()
    d.free()
    ctx.trim()
    # the record exchange at full size, eight "ranks" in this process: every rank's records from its own shard (+ halo),
    # every owner's count over its buckets' pieces; the owners' digests add up to the single-GPU digest
    import importlib
    sh = importlib.import_module(pkg.__name__ + ".shard_math")
    rows = n - k + 1
    nb = ctx.sk_buckets(rows, k)
    recs = []
    for first, cnt, lo, hi in sh.shard_ranges(n, k, 8):
        ds = ctx.synth(seed + lo // 32, hi - lo)
        recs.append(ctx.sk_records(ds, k, 0, cnt, rows))
        ds.free()
    t_sum = d_sum = u_sum = c_sum = 0
    for lo_b, hi_b in sh.bucket_owner_ranges(nb, 8):
        pieces = [(r.device_ptr + 16 * int(r.offsets[b]), int(r.offsets[b + 1] - r.offsets[b]), b) for r in recs for b in range(lo_b, hi_b)]
        ho = ctx.count_records(pieces, k, rows)
        t, dd, u, c = ho.summary()
        assert ho.total == t
        t_sum, d_sum, u_sum, c_sum = t_sum + t, d_sum + dd, u_sum + u, (c_sum + c) & M
        ho.free()
    for r in recs:
        r.free()
    assert (t_sum, d_sum, u_sum, c_sum) == oracle_digest("4")
    ctx.trim()
