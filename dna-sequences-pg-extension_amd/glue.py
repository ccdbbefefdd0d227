"""ctypes binding of libdna_glue.so (glue/dna_glue.h): the reference's SQL-visible functions on the
k-mer path, spelled the way test.sql spells them.  An ereport(ERROR) becomes GlueError(text)."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


class GlueError(Exception):
    pass


class _Kmer(C.Structure):
    _fields_ = [("length", C.c_int32), ("bit_sequence", C.c_uint64)]


class _Qkmer(C.Structure):
    _fields_ = [("sequence", C.c_char * 33)]


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "libdna_glue.so")
        if not os.path.exists(path):
            raise ImportError(f"{path} is missing: run __graft_entry__.build()")
        L = C.CDLL(path)
        vp = C.c_void_p
        L.dna_glue_errmsg.restype = C.c_char_p
        L.dna_in.restype = vp
        L.dna_in.argtypes = [C.c_char_p]
        L.dna_out.restype = vp
        L.dna_out.argtypes = [vp]
        L.dna_free.argtypes = [vp]
        L.dna_length.restype = C.c_uint64
        L.dna_length.argtypes = [vp]
        L.kmer_in.restype = C.c_bool
        L.kmer_in.argtypes = [C.c_char_p, C.POINTER(_Kmer)]
        L.kmer_out.restype = vp
        L.kmer_out.argtypes = [C.POINTER(_Kmer)]
        L.qkmer_in.restype = C.c_bool
        L.qkmer_in.argtypes = [C.c_char_p, C.POINTER(_Qkmer)]
        L.kmer_eq.restype = C.c_bool
        L.kmer_eq.argtypes = [C.POINTER(_Kmer)] * 2
        L.kmer_ne.restype = C.c_bool
        L.kmer_ne.argtypes = [C.POINTER(_Kmer)] * 2
        L.kmer_hash.restype = C.c_int32
        L.kmer_hash.argtypes = [C.POINTER(_Kmer)]
        L.starts_with.argtypes = [C.POINTER(_Kmer)] * 2
        L.contains.argtypes = [C.POINTER(_Qkmer), C.POINTER(_Kmer)]
        L.generate_kmers_begin.restype = vp
        L.generate_kmers_begin.argtypes = [vp, C.c_int]
        L.generate_kmers_where_begin.restype = vp
        L.generate_kmers_where_begin.argtypes = [vp, C.c_int, C.c_char, C.POINTER(_Kmer), C.POINTER(_Qkmer)]
        L.generate_kmers_next.restype = C.c_bool
        L.generate_kmers_next.argtypes = [vp, C.POINTER(_Kmer)]
        L.generate_kmers_failed.restype = C.c_bool
        L.generate_kmers_failed.argtypes = [vp]
        L.generate_kmers_end.argtypes = [vp]
        L.count_kmers_begin.restype = vp
        L.count_kmers_begin.argtypes = [vp, C.c_int]
        L.count_kmers_next.restype = C.c_bool
        L.count_kmers_next.argtypes = [vp, C.POINTER(_Kmer), C.POINTER(C.c_int64)]
        L.count_kmers_totals.argtypes = [vp] + [C.POINTER(C.c_int64)] * 3
        L.count_kmers_end.argtypes = [vp]
        L.dna_send.restype = vp
        L.dna_send.argtypes = [vp, C.POINTER(C.c_size_t)]
        L.dna_recv.restype = vp
        L.dna_recv.argtypes = [C.c_char_p, C.c_size_t]
        L.kmer_send.restype = None
        L.kmer_send.argtypes = [C.POINTER(_Kmer), C.c_char_p]
        L.kmer_recv.restype = C.c_bool
        L.kmer_recv.argtypes = [C.c_char_p, C.POINTER(_Kmer)]
        L.dna_glue_shutdown.restype = None
        L.dna_glue_set_gpus.restype = None
        L.dna_glue_set_gpus.argtypes = [C.c_int, C.POINTER(C.c_int), C.c_int]
        _LIB = L
    return _LIB


_libc = C.CDLL(None)
_libc.free.argtypes = [C.c_void_p]


def _err():
    return GlueError(lib().dna_glue_errmsg().decode())


def _take_str(p):
    if not p:
        raise _err()
    s = C.string_at(p).decode()
    _libc.free(p)
    return s


class dna:
    """the `dna` type: dna('ATCG') is dna_in"""

    def __init__(self, text=None, p=None):
        self.p = p if p is not None else lib().dna_in(text.encode())
        if not self.p:
            raise _err()

    def __str__(self):
        return _take_str(lib().dna_out(self.p))

    def send(self):
        """dna_send: the binary wire image"""
        n = C.c_size_t()
        buf = lib().dna_send(self.p, C.byref(n))
        if not buf:
            raise _err()
        raw = C.string_at(buf, n.value)
        _libc.free(buf)
        return raw

    @classmethod
    def recv(cls, wire):
        """dna_recv"""
        b = bytes(wire)
        p = lib().dna_recv(b, len(b))
        if not p:
            raise _err()
        return cls(p=p)

    def __len__(self):
        return int(lib().dna_length(self.p))

    def __del__(self):
        if getattr(self, "p", None) and _LIB is not None:
            _LIB.dna_free(self.p)
            self.p = None


class kmer:
    def __init__(self, text=None, c=None):
        self.c = c if c is not None else _Kmer()
        if text is not None and not lib().kmer_in(text.encode(), C.byref(self.c)):
            raise _err()

    def __str__(self):
        return _take_str(lib().kmer_out(C.byref(self.c)))

    def __eq__(self, other):                       # kmer = kmer
        return bool(lib().kmer_eq(C.byref(self.c), C.byref(other.c)))

    def __ne__(self, other):
        return bool(lib().kmer_ne(C.byref(self.c), C.byref(other.c)))

    def __hash__(self):
        return kmer_hash(self)

    def send(self):
        buf = C.create_string_buffer(12)
        lib().kmer_send(C.byref(self.c), buf)
        return buf.raw

    @classmethod
    def recv(cls, wire):
        c = _Kmer()
        if not lib().kmer_recv(bytes(wire), C.byref(c)):
            raise _err()
        return cls(c=c)


class qkmer:
    def __init__(self, text):
        self.c = _Qkmer()
        if not lib().qkmer_in(text.encode(), C.byref(self.c)):
            raise _err()

    def __str__(self):
        return self.c.sequence.decode()


def kmer_hash(k):
    return int(lib().kmer_hash(C.byref(k.c)))


def starts_with(k, prefix):                        # k ^@ prefix
    r = lib().starts_with(C.byref(k.c), C.byref(prefix.c))
    if r < 0:
        raise _err()
    return bool(r)


def contains(pattern, k):                          # pattern @> k
    r = lib().contains(C.byref(pattern.c), C.byref(k.c))
    if r < 0:
        raise _err()
    return bool(r)


def _drain(g):
    if not g:
        raise _err()
    rows, c = [], _Kmer()
    while lib().generate_kmers_next(g, C.byref(c)):
        rows.append(kmer(c=_Kmer(c.length, c.bit_sequence)))
    failed = lib().generate_kmers_failed(g)
    lib().generate_kmers_end(g)
    if failed:
        raise _err()
    return rows


def generate_kmers(d, k):
    """SELECT generate_kmers(d, k)"""
    if isinstance(d, str):
        d = dna(d)                                 # the implicit text -> dna cast of test.sql:46-47
    return _drain(lib().generate_kmers_begin(d.p, k))


def generate_kmers_where(d, k, op, rhs):
    """SELECT k.kmer FROM generate_kmers(d, k) AS k(kmer) WHERE <op>, op in '=', '^@', '@>'"""
    if isinstance(d, str):
        d = dna(d)
    if op == "@>":
        return _drain(lib().generate_kmers_where_begin(d.p, k, b"@", None, C.byref(rhs.c)))
    return _drain(lib().generate_kmers_where_begin(d.p, k, b"=" if op == "=" else b"^", C.byref(rhs.c), None))


def set_gpus(devices, transport=0):
    """count_kmers shards over these HIP devices from now on (dna_glue_set_gpus); [0] = single GPU"""
    n = len(devices)
    lib().dna_glue_set_gpus(n, (C.c_int * n)(*devices), transport)


def count_kmers(d, k):
    """SELECT k.kmer, count(*) FROM generate_kmers(d, k) AS k(kmer) GROUP BY k.kmer
    -> ([(kmer, count)...], (total, distinct, unique))"""
    if isinstance(d, str):
        d = dna(d)
    c = lib().count_kmers_begin(d.p, k)
    if not c:
        raise _err()
    rows, km, cnt = [], _Kmer(), C.c_int64()
    while lib().count_kmers_next(c, C.byref(km), C.byref(cnt)):
        rows.append((kmer(c=_Kmer(km.length, km.bit_sequence)), cnt.value))
    t, dd, u = C.c_int64(), C.c_int64(), C.c_int64()
    lib().count_kmers_totals(c, C.byref(t), C.byref(dd), C.byref(u))
    lib().count_kmers_end(c)
    return rows, (t.value, dd.value, u.value)
