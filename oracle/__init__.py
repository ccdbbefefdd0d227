"""ctypes binding of the CPU oracle (oracle/kmer_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg -- as the checker or the timed CPU baseline, never by the product package.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

u64p = C.POINTER(C.c_uint64)


def build(force=False):
    so = os.path.join(_HERE, "liboracle.so")
    src = [os.path.join(_HERE, f) for f in ("kmer_oracle.c", "kmer_oracle.h")]
    if force or not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in src):
        subprocess.check_call(["make", "-s", "-C", _HERE, "liboracle.so"])
    return so


def lib():
    global _LIB
    if _LIB is None:
        L = C.CDLL(build())
        L.orc_strerror.restype = C.c_char_p
        L.orc_strerror.argtypes = [C.c_int]
        L.orc_dna_num_words.restype = C.c_uint64
        L.orc_dna_num_words.argtypes = [C.c_uint64]
        L.orc_dna_encode.argtypes = [C.c_char_p, u64p]
        L.orc_dna_decode.argtypes = [u64p, C.c_uint64, C.c_char_p]
        L.orc_kmer_encode.argtypes = [C.c_char_p, C.POINTER(C.c_int32), u64p]
        L.orc_kmer_decode.argtypes = [C.c_uint64, C.c_int, C.c_char_p]
        L.orc_generate_kmers_count.argtypes = [C.c_uint64, C.c_int, u64p]
        for f in (L.orc_generate_kmers, L.orc_generate_kmers_fast):
            f.argtypes = [u64p, C.c_uint64, C.c_int, C.c_uint64, C.c_uint64, u64p]
        L.orc_table_kmers_count.argtypes = [u64p, C.c_uint64, C.c_int, u64p]
        L.orc_generate_kmers_table.argtypes = [u64p, C.c_uint64, u64p, C.c_uint64, C.c_int, C.c_int, u64p]
        L.orc_kmer_eq.argtypes = [C.c_int32, C.c_uint64, C.c_int32, C.c_uint64]
        L.orc_kmer_hash.restype = C.c_uint32
        L.orc_kmer_hash.argtypes = [C.c_uint64]
        L.orc_pg_hash_uint32.restype = C.c_uint32
        L.orc_pg_hash_uint32.argtypes = [C.c_uint32]
        L.orc_starts_with.argtypes = [C.c_int32, C.c_uint64, C.c_int32, C.c_uint64, C.POINTER(C.c_int)]
        L.orc_qkmer_validate.argtypes = [C.c_char_p]
        L.orc_contains.argtypes = [C.c_char_p, C.c_int32, C.c_uint64, C.POINTER(C.c_int)]
        L.orc_generate_kmers_contains.argtypes = [u64p, C.c_uint64, C.c_int, C.c_char_p, u64p, u64p,
                                                  C.c_uint64, u64p]
        for f in (L.orc_generate_kmers_starts_with, L.orc_generate_kmers_equals):
            f.argtypes = [u64p, C.c_uint64, C.c_int, C.c_int32, C.c_uint64, u64p, u64p, C.c_uint64, u64p]
        L.orc_count_keys.argtypes = [u64p, C.c_uint64, C.POINTER(u64p), C.POINTER(u64p), u64p]
        L.orc_count_kmers.argtypes = [u64p, C.c_uint64, C.c_int, C.c_int, C.POINTER(u64p),
                                      C.POINTER(u64p), u64p]
        L.orc_count_kmers_slice.argtypes = [u64p, C.c_uint64, C.c_int, C.c_uint32, C.c_uint32, C.POINTER(u64p),
                                            C.POINTER(u64p), u64p]
        L.orc_free.argtypes = [C.c_void_p]
        L.orc_hist_summary.argtypes = [u64p, u64p, C.c_uint64, u64p, u64p, u64p]
        L.orc_pair_mix.restype = C.c_uint64
        L.orc_pair_mix.argtypes = [C.c_uint64, C.c_uint64]
        L.orc_splitmix64.restype = C.c_uint64
        L.orc_splitmix64.argtypes = [C.c_uint64]
        L.orc_synth_words.argtypes = [C.c_uint64, C.c_uint64, u64p]
        L.orc_synth_words_repeat.argtypes = [C.c_uint64, C.c_uint64, C.c_uint64, u64p]
        L.orc_dna_wire_size.restype = C.c_uint64
        L.orc_dna_wire_size.argtypes = [C.c_uint64]
        L.orc_dna_to_wire.argtypes = [u64p, C.c_uint64, C.c_void_p]
        L.orc_dna_to_wire.restype = None
        L.orc_dna_from_wire.argtypes = [C.c_void_p, C.c_uint64, u64p, u64p]
        L.orc_kmer_to_wire.argtypes = [C.c_int32, C.c_uint64, C.c_void_p]
        L.orc_kmer_to_wire.restype = None
        L.orc_kmer_from_wire.argtypes = [C.c_void_p, C.POINTER(C.c_int32), u64p]
        _LIB = L
    return _LIB


class OracleError(Exception):
    def __init__(self, code):
        self.code = code
        super().__init__(lib().orc_strerror(code).decode())


def _chk(rc):
    if rc:
        raise OracleError(rc)


def _p(a):
    return a.ctypes.data_as(u64p)


def num_words(n_bases):
    return int(lib().orc_dna_num_words(n_bases))


def dna_encode(seq):
    """text -> (words uint64[], n_bases); dna.c:178-202"""
    b = seq.encode()
    words = np.zeros(max(num_words(len(b)), 1), dtype=np.uint64)
    _chk(lib().orc_dna_encode(b, _p(words)))
    return words[:num_words(len(b))], len(b)


def dna_decode(words, n_bases):
    buf = C.create_string_buffer(n_bases + 1)
    w = np.ascontiguousarray(words, dtype=np.uint64)
    lib().orc_dna_decode(_p(w), n_bases, buf)
    return buf.value.decode()


def kmer_encode(seq):
    ln, bits = C.c_int32(), C.c_uint64()
    _chk(lib().orc_kmer_encode(seq.encode(), C.byref(ln), C.byref(bits)))
    return ln.value, bits.value


def kmer_decode(bits, k):
    buf = C.create_string_buffer(k + 1)
    lib().orc_kmer_decode(int(bits), k, buf)
    return buf.value.decode()


def n_kmers(n_bases, k):
    out = C.c_uint64()
    _chk(lib().orc_generate_kmers_count(n_bases, k, C.byref(out)))
    return out.value


def generate_kmers(words, n_bases, k, first=0, count=None, faithful=True):
    total = n_kmers(n_bases, k)
    if count is None:
        count = max(total - first, 0)
    count = min(count, max(total - first, 0))
    out = np.empty(count, dtype=np.uint64)
    w = np.ascontiguousarray(words, dtype=np.uint64)
    fn = lib().orc_generate_kmers if faithful else lib().orc_generate_kmers_fast
    _chk(fn(_p(w), n_bases, k, first, count, _p(out)))
    return out


def generate_kmers_table(words, starts, k, faithful=False):
    """rows of generate_kmers(seq, k) for every sequence of a table, in table order (test.sql:140-150): `words` = the
    concatenated packed stream, starts[i] = first base of sequence i, starts[-1] = all bases"""
    st = np.ascontiguousarray(starts, dtype=np.uint64)
    n_seqs = len(st) - 1
    rows = C.c_uint64()
    _chk(lib().orc_table_kmers_count(_p(st), n_seqs, k, C.byref(rows)))
    out = np.empty(rows.value, dtype=np.uint64)
    w = np.ascontiguousarray(words, dtype=np.uint64)
    _chk(lib().orc_generate_kmers_table(_p(w), int(st[-1]), _p(st), n_seqs, k, 1 if faithful else 0, _p(out)))
    return out


def kmer_hash(bits):
    """signed int4 as SQL shows it (dna--1.0.sql:204-207 RETURNS INTEGER)"""
    h = lib().orc_kmer_hash(int(bits))
    return h - (1 << 32) if h >= (1 << 31) else h


def pg_hashint4(k):
    h = lib().orc_pg_hash_uint32(k & 0xFFFFFFFF)
    return h - (1 << 32) if h >= (1 << 31) else h


def starts_with(klen, kbits, plen, pbits):
    r = C.c_int()
    _chk(lib().orc_starts_with(klen, int(kbits), plen, int(pbits), C.byref(r)))
    return bool(r.value)


def qkmer_validate(pattern):
    _chk(lib().orc_qkmer_validate(pattern.encode()))


def contains(pattern, klen, kbits):
    r = C.c_int()
    _chk(lib().orc_contains(pattern.encode(), klen, int(kbits), C.byref(r)))
    return bool(r.value)


def _filtered(fn, words, n_bases, k, *args):
    w = np.ascontiguousarray(words, dtype=np.uint64)
    cap = n_kmers(n_bases, k)
    keys = np.empty(max(cap, 1), dtype=np.uint64)
    pos = np.empty(max(cap, 1), dtype=np.uint64)
    n = C.c_uint64()
    _chk(fn(_p(w), n_bases, k, *args, _p(keys), _p(pos), cap, C.byref(n)))
    return keys[:n.value].copy(), pos[:n.value].copy()


def generate_kmers_contains(words, n_bases, k, pattern):
    return _filtered(lib().orc_generate_kmers_contains, words, n_bases, k, pattern.encode())


def generate_kmers_starts_with(words, n_bases, k, plen, pbits):
    return _filtered(lib().orc_generate_kmers_starts_with, words, n_bases, k, plen, int(pbits))


def generate_kmers_equals(words, n_bases, k, qlen, qbits):
    return _filtered(lib().orc_generate_kmers_equals, words, n_bases, k, qlen, int(qbits))


def _take(pk, pc, d):
    keys = np.ctypeslib.as_array(pk, shape=(max(d, 1),))[:d].copy()
    counts = np.ctypeslib.as_array(pc, shape=(max(d, 1),))[:d].copy()
    lib().orc_free(pk)
    lib().orc_free(pc)
    return keys, counts


def count_keys(keys):
    k = np.ascontiguousarray(keys, dtype=np.uint64)
    pk, pc, d = u64p(), u64p(), C.c_uint64()
    _chk(lib().orc_count_keys(_p(k), k.size, C.byref(pk), C.byref(pc), C.byref(d)))
    return _take(pk, pc, d.value)


def count_kmers(words, n_bases, k, faithful=False):
    """GROUP BY kmer, count(*): (keys ascending, counts)"""
    w = np.ascontiguousarray(words, dtype=np.uint64)
    pk, pc, d = u64p(), u64p(), C.c_uint64()
    _chk(lib().orc_count_kmers(_p(w), n_bases, k, int(faithful), C.byref(pk), C.byref(pc), C.byref(d)))
    return _take(pk, pc, d.value)


def count_kmers_slice(words, n_bases, k, slice_, n_slices):
    """the groups of one slice of the key space (disjoint between slices, union = count_kmers)"""
    w = np.ascontiguousarray(words, dtype=np.uint64)
    pk, pc, d = u64p(), u64p(), C.c_uint64()
    _chk(lib().orc_count_kmers_slice(_p(w), n_bases, k, slice_, n_slices, C.byref(pk), C.byref(pc), C.byref(d)))
    return _take(pk, pc, d.value)


def hist_summary(keys, counts):
    """(total, distinct, unique, checksum) -- test.sql:107-119 + order-independent digest"""
    k = np.ascontiguousarray(keys, dtype=np.uint64)
    c = np.ascontiguousarray(counts, dtype=np.uint64)
    t, u, s = C.c_uint64(), C.c_uint64(), C.c_uint64()
    lib().orc_hist_summary(_p(k), _p(c), k.size, C.byref(t), C.byref(u), C.byref(s))
    return t.value, int(k.size), u.value, s.value


def synth_words(seed, n_bases):
    words = np.empty(max(num_words(n_bases), 1), dtype=np.uint64)
    lib().orc_synth_words(seed, n_bases, _p(words))
    return words[:num_words(n_bases)]


def synth_words_repeat(seed, n_bases, motif_len):
    words = np.empty(max(num_words(n_bases), 1), dtype=np.uint64)
    lib().orc_synth_words_repeat(seed, n_bases, motif_len, _p(words))
    return words[:num_words(n_bases)]


def dna_to_wire(words, n_bases):
    """dna_send (dna.c:270-291) with a working int64 length: bytes"""
    w = np.ascontiguousarray(words, dtype=np.uint64)
    buf = C.create_string_buffer(int(lib().orc_dna_wire_size(n_bases)))
    lib().orc_dna_to_wire(_p(w), n_bases, buf)
    return buf.raw


def dna_from_wire(wire):
    """dna_recv (dna.c:244-268): bytes -> (words, n_bases)"""
    b = bytes(wire)
    n = C.c_uint64()
    _chk(lib().orc_dna_from_wire(b, len(b), C.byref(n), None))
    w = np.zeros(num_words(n.value), dtype=np.uint64)
    _chk(lib().orc_dna_from_wire(b, len(b), C.byref(n), _p(w)))
    return w, n.value


def kmer_to_wire(length, bits):
    buf = C.create_string_buffer(12)
    lib().orc_kmer_to_wire(length, bits, buf)
    return buf.raw


def kmer_from_wire(wire):
    ln, bits = C.c_int32(), C.c_uint64()
    _chk(lib().orc_kmer_from_wire(bytes(wire), C.byref(ln), C.byref(bits)))
    return ln.value, bits.value
