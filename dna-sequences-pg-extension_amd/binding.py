"""ctypes binding of include/dnagpu.h (libdnagpu.so).  One method per C entry point; numpy arrays
in and out; errors raise DnaGpuError carrying the status code and the reference's message text."""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

OK = 0
ERR_INVALID_K = 1
ERR_QKMER_LEN_MISMATCH = 2
ERR_PREFIX_TOO_LONG = 3
ERR_QKMER_INVALID = 4
ERR_BAD_ARG = 5
ERR_TOO_LARGE = 6
ERR_NO_DEVICE = 7
ERR_OOM = 8
ERR_HIP = 9
ERR_DNA_EMPTY = 11
ERR_DNA_INVALID_CHAR = 12
DEBUG_POISON_POOL = 1
DEBUG_FORCE_SUPERKMER = 2
DEBUG_HEAVY_EXPAND = 4
DEBUG_GUARD_POOL = 8
DEBUG_NO_SPEC1 = 16
DEBUG_SPEC1_OVERFLOW = 32
DEBUG_SLAB0 = 64
DEBUG_NO_SLAB0 = 128
DEBUG_SLAB0_OVERFLOW = 256
DEBUG_SAMPLE1 = 512                   # level 1: regions from a sampled histogram whatever the coarse buckets look like


def _env_debug():
    """test-harness switches (this wrapper is test / bench tooling): DNAGPU_TEST_POISON=1 runs every context with
    DNAGPU_DEBUG_POISON_POOL (all work buffers pre-filled with 0xFF), DNAGPU_TEST_GUARD=1 with DNAGPU_DEBUG_GUARD_POOL
    (a guard band behind every work buffer, checked whenever a histogram, a sequence or a buffer is freed)"""
    f = 0
    if os.environ.get("DNAGPU_TEST_POISON", "0") not in ("", "0"):
        f |= DEBUG_POISON_POOL
    if os.environ.get("DNAGPU_TEST_GUARD", "0") not in ("", "0"):
        f |= DEBUG_GUARD_POOL
    return f

FILTER_EQUALS = 1
FILTER_STARTS_WITH = 2
FILTER_CONTAINS = 3

MAX_PHASES = 48

u64p = C.POINTER(C.c_uint64)


class _FilterC(C.Structure):
    _fields_ = [("kind", C.c_int32), ("length", C.c_int32), ("bits", C.c_uint64),
                ("pattern", C.c_char * 36), ("reserved", C.c_int32)]


class _PhaseTimes(C.Structure):
    _fields_ = [("n", C.c_int), ("names", C.c_char_p * MAX_PHASES), ("ms", C.c_float * MAX_PHASES)]


class _MultiTimes(C.Structure):
    _fields_ = [("records_ms", C.c_double), ("exchange_ms", C.c_double), ("hidden_ms", C.c_double),
                ("count_ms", C.c_double), ("total_ms", C.c_double), ("bytes_moved", C.c_uint64), ("parts", C.c_int)]


def lib_path():
    # DNAGPU_LIB_PATH: A/B runs of two builds on one box (tools/ab.sh); this wrapper is test / bench tooling, the
    # library itself reads no environment variable
    return os.environ.get("DNAGPU_LIB_PATH") or os.path.join(_HERE, "libdnagpu.so")


def lib():
    """Loads libdnagpu.so; raises (never falls back) when it has not been built."""
    global _LIB
    if _LIB is not None:
        return _LIB
    path = lib_path()
    if not os.path.exists(path):
        raise ImportError(f"{path} is missing: run __graft_entry__.build() (hipcc --offload-arch=gfx950)")
    L = C.CDLL(path)
    vp = C.c_void_p
    L.dnagpu_strerror.restype = C.c_char_p
    L.dnagpu_strerror.argtypes = [C.c_int]
    L.dnagpu_last_error.restype = C.c_char_p
    L.dnagpu_abi_version.restype = C.c_int
    L.dnagpu_init.argtypes = [C.c_int, C.POINTER(vp)]
    L.dnagpu_destroy.argtypes = [vp]
    L.dnagpu_destroy.restype = None
    L.dnagpu_synchronize.argtypes = [vp]
    L.dnagpu_stream.argtypes = [vp]
    L.dnagpu_stream.restype = vp
    L.dnagpu_trim.argtypes = [vp]
    L.dnagpu_device_bytes.argtypes = [vp]
    L.dnagpu_device_bytes.restype = C.c_uint64
    L.dnagpu_dna_upload.argtypes = [vp, u64p, C.c_uint64, C.POINTER(vp)]
    L.dnagpu_dna_wrap.argtypes = [vp, vp, C.c_uint64, C.c_uint64, C.POINTER(vp)]
    L.dnagpu_dna_synth.argtypes = [vp, C.c_uint64, C.c_uint64, C.c_uint64, C.POINTER(vp)]
    L.dnagpu_dna_download.argtypes = [vp, vp, u64p]
    L.dnagpu_dna_pack.argtypes = [vp, C.c_char_p, C.c_uint64, C.c_int, C.POINTER(vp), u64p, C.c_char_p]
    L.dnagpu_dna_unpack.argtypes = [vp, vp, C.c_uint64, C.c_uint64, vp, C.c_int]
    L.dnagpu_kmers_to_text.argtypes = [vp, vp, C.c_uint64, C.c_int, vp, C.c_int]
    L.dnagpu_dna_wire_size.argtypes = [C.c_uint64]
    L.dnagpu_dna_wire_size.restype = C.c_uint64
    L.dnagpu_dna_from_wire.argtypes = [vp, vp, C.c_uint64, C.c_int, C.POINTER(vp)]
    L.dnagpu_dna_to_wire.argtypes = [vp, vp, vp, C.c_uint64, C.c_int]
    L.dnagpu_dna_length.argtypes = [vp]
    L.dnagpu_dna_length.restype = C.c_uint64
    L.dnagpu_dna_device_words.argtypes = [vp]
    L.dnagpu_dna_device_words.restype = vp
    L.dnagpu_dna_free.argtypes = [vp, vp]
    L.dnagpu_dna_free.restype = None
    L.dnagpu_kmer_count.argtypes = [C.c_uint64, C.c_int, u64p]
    L.dnagpu_generate_kmers.argtypes = [vp, vp, C.c_int, C.c_uint64, C.c_uint64, vp, C.c_int]
    L.dnagpu_generate_kmers_filtered.argtypes = [vp, vp, C.c_int, C.POINTER(_FilterC), C.c_uint64,
                                                 C.c_uint64, vp, vp, C.c_uint64, u64p, C.c_int]
    L.dnagpu_count_kmers.argtypes = [vp, vp, C.c_int, C.c_uint64, C.c_uint64, C.POINTER(vp)]
    L.dnagpu_count_kmers_unordered.argtypes = [vp, vp, C.c_int, C.c_uint64, C.c_uint64, C.POINTER(vp)]
    L.dnagpu_hist_is_sorted.argtypes = [vp]
    L.dnagpu_count_kmers_batch.argtypes = [vp, vp, u64p, C.c_uint64, C.c_int, C.POINTER(vp)]
    L.dnagpu_dna_set_sequences.argtypes = [vp, vp, u64p, C.c_uint64]
    L.dnagpu_dna_sequences.argtypes = [vp]
    L.dnagpu_dna_sequences.restype = C.c_uint64
    L.dnagpu_count_kmers_table.argtypes = [vp, vp, C.c_int, C.POINTER(vp)]
    L.dnagpu_sk_buckets.argtypes = [vp, C.c_uint64, C.c_int]
    L.dnagpu_sk_records.argtypes = [vp, vp, C.c_int, C.c_uint64, C.c_uint64, C.c_uint64, C.POINTER(vp)]
    L.dnagpu_records_buckets.argtypes = [vp]
    L.dnagpu_records_buckets.restype = C.c_uint32
    L.dnagpu_records_offsets.argtypes = [vp, u64p]
    L.dnagpu_records_device.argtypes = [vp]
    L.dnagpu_records_device.restype = vp
    L.dnagpu_records_free.argtypes = [vp, vp]
    L.dnagpu_records_free.restype = None
    L.dnagpu_count_records.argtypes = [vp, C.POINTER(vp), u64p, C.POINTER(C.c_uint32), C.c_uint32, C.c_int, C.c_uint64,
                                       C.POINTER(vp)]
    L.dnagpu_count_keys.argtypes = [vp, vp, C.c_uint64, C.c_int, C.POINTER(vp)]
    L.dnagpu_count_kmers_owned.argtypes = [vp, vp, C.c_int, C.c_uint64, C.c_uint64, C.c_int, C.c_int, C.POINTER(vp)]
    L.dnagpu_count_keys_in_range.argtypes = [vp, vp, C.c_uint64, C.c_int, C.c_uint64, C.c_uint64, C.POINTER(vp)]
    L.dnagpu_hist_distinct.argtypes = [vp]
    L.dnagpu_hist_distinct.restype = C.c_uint64
    L.dnagpu_hist_total.argtypes = [vp]
    L.dnagpu_hist_total.restype = C.c_uint64
    L.dnagpu_hist_extent.argtypes = [vp]
    L.dnagpu_hist_extent.restype = C.c_uint64
    L.dnagpu_hist_device_keys.argtypes = [vp]
    L.dnagpu_hist_device_keys.restype = vp
    L.dnagpu_hist_device_counts.argtypes = [vp]
    L.dnagpu_hist_device_counts.restype = vp
    L.dnagpu_hist_download.argtypes = [vp, vp, C.c_uint64, C.c_uint64, u64p, u64p]
    L.dnagpu_hist_summary.argtypes = [vp, vp, u64p, u64p, u64p]
    L.dnagpu_hist_sorted_view.argtypes = [vp, vp, C.c_uint64, C.c_uint64, vp, vp]
    L.dnagpu_hist_merge.argtypes = [vp, vp, vp, C.POINTER(vp)]
    L.dnagpu_hist_free.argtypes = [vp, vp]
    L.dnagpu_hist_free.restype = None
    L.dnagpu_partition_kmers.argtypes = [vp, vp, C.c_int, C.c_uint64, C.c_uint64, C.c_int, C.POINTER(vp), u64p]
    L.dnagpu_buffer_alloc.argtypes = [vp, C.c_uint64, C.POINTER(vp)]
    L.dnagpu_buffer_free.argtypes = [vp, vp]
    L.dnagpu_buffer_free.restype = None
    L.dnagpu_buffer_download.argtypes = [vp, vp, C.c_uint64, vp]
    L.dnagpu_buffer_upload.argtypes = [vp, vp, vp, C.c_uint64]
    L.dnagpu_kmer_hash.argtypes = [vp, vp, C.c_uint64, vp, C.c_int]
    L.dnagpu_kmer_match.argtypes = [vp, vp, C.c_uint64, C.c_int, C.POINTER(_FilterC), vp, C.c_int]
    L.dnagpu_multi_init.argtypes = [C.POINTER(C.c_int), C.c_int, C.c_int, C.POINTER(vp)]
    L.dnagpu_multi_destroy.argtypes = [vp]
    L.dnagpu_multi_destroy.restype = None
    L.dnagpu_multi_size.argtypes = [vp]
    L.dnagpu_multi_ctx.argtypes = [vp, C.c_int]
    L.dnagpu_multi_ctx.restype = vp
    L.dnagpu_multi_transport.argtypes = [vp]
    L.dnagpu_multi_transport.restype = C.c_char_p
    L.dnagpu_multi_dna_upload.argtypes = [vp, u64p, C.c_uint64, C.POINTER(vp)]
    L.dnagpu_multi_dna_synth.argtypes = [vp, C.c_uint64, C.c_uint64, C.c_uint64, C.POINTER(vp)]
    L.dnagpu_multi_dna_length.argtypes = [vp]
    L.dnagpu_multi_dna_length.restype = C.c_uint64
    L.dnagpu_multi_dna_free.argtypes = [vp, vp]
    L.dnagpu_multi_dna_free.restype = None
    L.dnagpu_count_multi.argtypes = [vp, vp, C.c_int, C.c_uint64, C.c_uint64, C.POINTER(vp)]
    L.dnagpu_count_multi_unordered.argtypes = [vp, vp, C.c_int, C.c_uint64, C.c_uint64, C.POINTER(vp)]
    L.dnagpu_device_count.restype = C.c_int
    L.dnagpu_multi_set_option.argtypes = [vp, C.c_int, C.c_double]
    L.dnagpu_multi_last_times.argtypes = [vp, C.POINTER(_MultiTimes)]
    L.dnagpu_multi_exchange_transport.argtypes = [vp]
    L.dnagpu_multi_exchange_transport.restype = C.c_char_p
    L.dnagpu_multi_rccl_ranks.argtypes = [vp]
    L.dnagpu_multi_last_phase_times.argtypes = [vp, C.c_int, C.POINTER(_PhaseTimes)]
    L.dnagpu_hist_parts.argtypes = [vp]
    L.dnagpu_hist_parts.restype = C.c_uint32
    L.dnagpu_hist_part.argtypes = [vp, C.c_uint32]
    L.dnagpu_hist_part.restype = vp
    L.dnagpu_last_phase_times.argtypes = [vp, C.POINTER(_PhaseTimes)]
    L.dnagpu_set_profiling.argtypes = [vp, C.c_int]
    L.dnagpu_set_debug.argtypes = [vp, C.c_uint]
    _LIB = L
    return L


class DnaGpuError(Exception):
    def __init__(self, code):
        self.code = code
        self.message = lib().dnagpu_strerror(code).decode()
        detail = lib().dnagpu_last_error().decode()
        super().__init__(self.message + (f" [{detail}]" if detail and code >= ERR_BAD_ARG else ""))


def _chk(rc):
    if rc != OK:
        raise DnaGpuError(rc)


def strerror(code):
    return lib().dnagpu_strerror(code).decode()


def device_count():
    """HIP devices this process can use (no context is created)"""
    return int(lib().dnagpu_device_count())


def abi_version():
    return lib().dnagpu_abi_version()


def kmer_count(n_bases, k):
    out = C.c_uint64()
    _chk(lib().dnagpu_kmer_count(n_bases, k, C.byref(out)))
    return out.value


class Filter:
    """Right-hand side of a WHERE operator on generate_kmers rows."""

    def __init__(self, kind, length=0, bits=0, pattern=""):
        self.c = _FilterC()
        self.c.kind = kind
        self.c.length = length
        self.c.bits = int(bits)
        self.c.pattern = pattern.encode()[:35]

    @staticmethod
    def equals(length, bits):          # kmer = q
        return Filter(FILTER_EQUALS, length, bits)

    @staticmethod
    def starts_with(length, bits):     # kmer ^@ prefix
        return Filter(FILTER_STARTS_WITH, length, bits)

    @staticmethod
    def contains(pattern):             # qkmer @> kmer
        return Filter(FILTER_CONTAINS, 0, 0, pattern)


class Dna:
    def __init__(self, ctx, handle):
        self.ctx, self.h = ctx, handle

    @property
    def n_bases(self):
        return int(lib().dnagpu_dna_length(self.h))

    @property
    def device_words(self):
        return lib().dnagpu_dna_device_words(self.h)

    def download(self):
        nw = (self.n_bases + 31) // 32
        out = np.empty(max(nw, 1), dtype=np.uint64)
        _chk(lib().dnagpu_dna_download(self.ctx.h, self.h, out.ctypes.data_as(u64p)))
        return out[:nw]

    def set_sequences(self, seq_starts):
        """this packed stream is a TABLE of sequences: seq_starts = the first base of each + the total (n_seqs + 1 entries),
        kept on the device beside it (dnagpu_dna_set_sequences) for Context.count_kmers_table"""
        st = np.ascontiguousarray(seq_starts, dtype=np.uint64)
        _chk(lib().dnagpu_dna_set_sequences(self.ctx.h, self.h, st.ctypes.data_as(u64p), max(len(st) - 1, 0)))

    @property
    def n_sequences(self):
        return int(lib().dnagpu_dna_sequences(self.h))

    def free(self):
        if self.h:
            lib().dnagpu_dna_free(self.ctx.h, self.h)
            self.h = None


class Records:
    """dnagpu_records: 16-byte super-k-mer records in device memory, grouped by coarse bucket"""

    def __init__(self, ctx, h):
        self.ctx, self.h = ctx, h
        self.n_buckets = int(lib().dnagpu_records_buckets(h))
        self.offsets = np.zeros(self.n_buckets + 1, dtype=np.uint64)
        _chk(lib().dnagpu_records_offsets(h, self.offsets.ctypes.data_as(u64p)))
        self.device_ptr = lib().dnagpu_records_device(h)
        self.n_records = int(self.offsets[-1])

    def free(self):
        if self.h:
            lib().dnagpu_records_free(self.ctx.h, self.h)
            self.h = None


class Hist:
    def __init__(self, ctx, handle):
        self.ctx, self.h = ctx, handle

    @property
    def distinct(self):
        return int(lib().dnagpu_hist_distinct(self.h))

    @property
    def total(self):
        return int(lib().dnagpu_hist_total(self.h))

    @property
    def extent(self):
        """slots of device_keys / device_counts in use (an unordered histogram may hold count-0 padding slots)"""
        return int(lib().dnagpu_hist_extent(self.h))

    @property
    def device_keys(self):
        return lib().dnagpu_hist_device_keys(self.h)

    @property
    def device_counts(self):
        return lib().dnagpu_hist_device_counts(self.h)

    @property
    def n_parts(self):
        """device-array sets this histogram consists of (the pipelined multi-GPU count makes several)"""
        return int(lib().dnagpu_hist_parts(self.h))

    def part_arrays(self, i):
        """(device keys pointer, device counts pointer, extent) of part i"""
        ph = C.c_void_p(lib().dnagpu_hist_part(self.h, i))
        return (lib().dnagpu_hist_device_keys(ph), lib().dnagpu_hist_device_counts(ph), int(lib().dnagpu_hist_extent(ph)))

    @property
    def is_sorted(self):
        return bool(lib().dnagpu_hist_is_sorted(self.h))

    def download(self, first=0, count=None):
        if count is None:
            count = self.distinct - first
        keys = np.empty(max(count, 1), dtype=np.uint64)
        counts = np.empty(max(count, 1), dtype=np.uint64)
        _chk(lib().dnagpu_hist_download(self.ctx.h, self.h, first, count, keys.ctypes.data_as(u64p),
                                        counts.ctypes.data_as(u64p)))
        return keys[:count], counts[:count]

    def sorted_view_device(self, dev_keys, dev_counts, first=0, count=None):
        """ascending-key groups [first, first+count) into caller-owned device arrays of uint64"""
        if count is None:
            count = self.distinct - first
        _chk(lib().dnagpu_hist_sorted_view(self.ctx.h, self.h, first, count, dev_keys, dev_counts))

    def summary(self):
        """(total, distinct, unique, checksum) -- same tuple as the oracle's hist_summary"""
        t, u, c = C.c_uint64(), C.c_uint64(), C.c_uint64()
        _chk(lib().dnagpu_hist_summary(self.ctx.h, self.h, C.byref(t), C.byref(u), C.byref(c)))
        return t.value, self.distinct, u.value, c.value

    def merge(self, other):
        """-> a new Hist: the groups of self and other added up (dnagpu_hist_merge)"""
        h = C.c_void_p()
        _chk(lib().dnagpu_hist_merge(self.ctx.h, self.h, other.h, C.byref(h)))
        return Hist(self.ctx, h)

    def free(self):
        if self.h:
            if self.ctx._base_debug & DEBUG_GUARD_POOL:
                self.ctx.synchronize()          # raises if a kernel wrote past the end of a work buffer
            lib().dnagpu_hist_free(self.ctx.h, self.h)
            self.h = None


class Context:
    def __init__(self, device=0):
        self.h = C.c_void_p()
        _chk(lib().dnagpu_init(device, C.byref(self.h)))
        # test harness switch (this wrapper is test/bench tooling): DNAGPU_TEST_POISON=1 runs every context
        # with DNAGPU_DEBUG_POISON_POOL, i.e. all work buffers pre-filled with 0xFF
        self._base_debug = _env_debug()
        if self._base_debug:
            self.set_debug(0)

    def set_debug(self, flags):
        """flags on top of what the environment asked for"""
        _chk(lib().dnagpu_set_debug(self.h, int(flags) | self._base_debug))

    def close(self):
        if self.h:
            lib().dnagpu_destroy(self.h)
            self.h = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def synchronize(self):
        _chk(lib().dnagpu_synchronize(self.h))

    @property
    def stream(self):
        return lib().dnagpu_stream(self.h)

    def trim(self):
        _chk(lib().dnagpu_trim(self.h))

    def device_bytes(self):
        return int(lib().dnagpu_device_bytes(self.h))

    def set_profiling(self, on):
        _chk(lib().dnagpu_set_profiling(self.h, int(bool(on))))

    def last_phase_times(self):
        pt = _PhaseTimes()
        _chk(lib().dnagpu_last_phase_times(self.h, C.byref(pt)))
        return [(pt.names[i].decode(), float(pt.ms[i])) for i in range(pt.n)]

    # ---- dna
    def upload(self, words, n_bases):
        w = np.ascontiguousarray(words, dtype=np.uint64)
        if w.size < (n_bases + 31) // 32:
            raise ValueError("words shorter than n_bases")
        h = C.c_void_p()
        _chk(lib().dnagpu_dna_upload(self.h, w.ctypes.data_as(u64p), n_bases, C.byref(h)))
        return Dna(self, h)

    def wrap(self, dev_ptr, n_words, n_bases):
        h = C.c_void_p()
        _chk(lib().dnagpu_dna_wrap(self.h, dev_ptr, n_words, n_bases, C.byref(h)))
        return Dna(self, h)

    def pack(self, text):
        """dna_in on the device: text (str/bytes of A/T/C/G) -> Dna"""
        b = text.encode() if isinstance(text, str) else bytes(text)
        h = C.c_void_p()
        pos, ch = C.c_uint64(), C.create_string_buffer(1)
        rc = lib().dnagpu_dna_pack(self.h, b, len(b), 0, C.byref(h), C.byref(pos), ch)
        if rc == ERR_DNA_INVALID_CHAR:
            e = DnaGpuError(rc)
            e.message = f"{e.message}: {ch.raw.decode(errors='replace')}"
            e.bad_pos = pos.value
            e.args = (e.message,)
            raise e
        _chk(rc)
        return Dna(self, h)

    def unpack(self, dna, first=0, count=None):
        if count is None:
            count = dna.n_bases - first
        buf = C.create_string_buffer(max(count, 1))
        _chk(lib().dnagpu_dna_unpack(self.h, dna.h, first, count, buf, 0))
        return buf.raw[:count].decode()

    def from_wire(self, wire):
        """dna_recv on the device: the binary wire image (bytes) -> Dna"""
        b = bytes(wire)
        h = C.c_void_p()
        _chk(lib().dnagpu_dna_from_wire(self.h, b, len(b), 0, C.byref(h)))
        return Dna(self, h)

    def to_wire(self, dna):
        """dna_send on the device: Dna -> the binary wire image (bytes)"""
        n = lib().dnagpu_dna_wire_size(dna.n_bases)
        buf = C.create_string_buffer(int(n))
        _chk(lib().dnagpu_dna_to_wire(self.h, dna.h, buf, n, 0))
        return buf.raw

    def kmers_to_text(self, keys, k):
        a = np.ascontiguousarray(keys, dtype=np.uint64)
        buf = C.create_string_buffer(max(a.size * (k + 1), 1))
        _chk(lib().dnagpu_kmers_to_text(self.h, a.ctypes.data, a.size, k, buf, 0))
        raw = buf.raw
        return [raw[i * (k + 1): i * (k + 1) + k].decode() for i in range(a.size)]

    def synth(self, seed, n_bases, motif_len=0):
        h = C.c_void_p()
        _chk(lib().dnagpu_dna_synth(self.h, seed, n_bases, motif_len, C.byref(h)))
        return Dna(self, h)

    # ---- generate_kmers
    def generate_kmers(self, dna, k, first=0, count=None):
        total = kmer_count(dna.n_bases, k)
        if count is None:
            count = max(total - first, 0)
        out = np.empty(max(count, 1), dtype=np.uint64)
        _chk(lib().dnagpu_generate_kmers(self.h, dna.h, k, first, count, out.ctypes.data, 0))
        return out[:count]

    def generate_kmers_device(self, dna, k, first, count, dev_out):
        _chk(lib().dnagpu_generate_kmers(self.h, dna.h, k, first, count, dev_out, 1))

    def generate_kmers_filtered(self, dna, k, flt, first=0, count=None, cap=None, want_keys=True,
                                want_pos=True):
        """-> (keys, positions, n_total_matches)"""
        total = kmer_count(dna.n_bases, k)
        if count is None:
            count = max(total - first, 0)
        if cap is None:
            cap = count
        keys = np.empty(max(cap, 1), dtype=np.uint64) if want_keys else None
        pos = np.empty(max(cap, 1), dtype=np.uint64) if want_pos else None
        n = C.c_uint64()
        _chk(lib().dnagpu_generate_kmers_filtered(
            self.h, dna.h, k, C.byref(flt.c), first, count,
            keys.ctypes.data if want_keys else None, pos.ctypes.data if want_pos else None,
            cap, C.byref(n), 0))
        m = min(n.value, cap)
        return (keys[:m] if want_keys else None, pos[:m] if want_pos else None, n.value)

    def count_matches_device(self, dna, k, flt, first, count, dev_keys=None, dev_pos=None, cap=0):
        n = C.c_uint64()
        _chk(lib().dnagpu_generate_kmers_filtered(self.h, dna.h, k, C.byref(flt.c), first, count,
                                                  dev_keys, dev_pos, cap, C.byref(n), 1))
        return n.value

    # ---- GROUP BY
    def count_kmers(self, dna, k, first=0, count=None):
        total = kmer_count(dna.n_bases, k)
        if count is None:
            count = max(total - first, 0)
        h = C.c_void_p()
        _chk(lib().dnagpu_count_kmers(self.h, dna.h, k, first, count, C.byref(h)))
        return Hist(self, h)

    def count_kmers_unordered(self, dna, k, first=0, count=None):
        """same groups, order unspecified (as PostgreSQL's GROUP BY): long k-mers go through super-k-mer partitioning"""
        total = kmer_count(dna.n_bases, k)
        if count is None:
            count = max(total - first, 0)
        h = C.c_void_p()
        _chk(lib().dnagpu_count_kmers_unordered(self.h, dna.h, k, first, count, C.byref(h)))
        return Hist(self, h)

    def count_kmers_batch(self, dna, seq_starts, k):
        """GROUP BY over a table of sequences (test.sql:140-150): dna = the sequences back to back, seq_starts = the first
        base of each + the total (n_seqs + 1 entries)"""
        st = np.ascontiguousarray(seq_starts, dtype=np.uint64)
        h = C.c_void_p()
        _chk(lib().dnagpu_count_kmers_batch(self.h, dna.h, st.ctypes.data_as(u64p), max(len(st) - 1, 0), k, C.byref(h)))
        return Hist(self, h)

    def count_kmers_table(self, dna, k):
        """the same over a table made resident with Dna.set_sequences: nothing crosses the bus per count"""
        h = C.c_void_p()
        _chk(lib().dnagpu_count_kmers_table(self.h, dna.h, k, C.byref(h)))
        return Hist(self, h)

    # ---- the unordered count in two halves (rows on several GPUs: sharded.count_sharded_exchange_records)
    def sk_buckets(self, global_rows, k):
        return int(lib().dnagpu_sk_buckets(self.h, global_rows, k))

    def sk_records(self, dna, k, first, count, global_rows):
        """super-k-mer records of rows [first, first+count), grouped by coarse bucket -> Records"""
        r = C.c_void_p()
        _chk(lib().dnagpu_sk_records(self.h, dna.h, k, first, count, global_rows, C.byref(r)))
        return Records(self, r)

    def count_records(self, pieces, k, global_rows):
        """pieces: [(device pointer, n_records, bucket)] -> Hist (unordered) of the k-mers in them"""
        n = len(pieces)
        ptrs = (C.c_void_p * max(n, 1))(*[C.c_void_p(int(p[0])) for p in pieces])
        lens = np.asarray([int(p[1]) for p in pieces], dtype=np.uint64)
        bks = np.asarray([int(p[2]) for p in pieces], dtype=np.uint32)
        h = C.c_void_p()
        _chk(lib().dnagpu_count_records(self.h, ptrs, lens.ctypes.data_as(u64p), bks.ctypes.data_as(C.POINTER(C.c_uint32)), n, k,
                                        global_rows, C.byref(h)))
        return Hist(self, h)

    def count_kmers_owned(self, dna, k, owner, n_owners, first=0, count=None):
        total = kmer_count(dna.n_bases, k)
        if count is None:
            count = max(total - first, 0)
        h = C.c_void_p()
        _chk(lib().dnagpu_count_kmers_owned(self.h, dna.h, k, first, count, owner, n_owners, C.byref(h)))
        return Hist(self, h)

    def count_keys_device(self, dev_keys, n, k):
        h = C.c_void_p()
        _chk(lib().dnagpu_count_keys(self.h, dev_keys, n, k, C.byref(h)))
        return Hist(self, h)

    def count_keys_device_in_range(self, dev_keys, n, k, key_min, key_max):
        h = C.c_void_p()
        _chk(lib().dnagpu_count_keys_in_range(self.h, dev_keys, n, k, key_min, key_max, C.byref(h)))
        return Hist(self, h)

    def partition_kmers(self, dna, k, first, count, n_owners):
        """-> (device pointer to `count` keys grouped by owner, offsets[n_owners+1])"""
        ptr = C.c_void_p()
        offs = np.zeros(n_owners + 1, dtype=np.uint64)
        _chk(lib().dnagpu_partition_kmers(self.h, dna.h, k, first, count, n_owners, C.byref(ptr),
                                          offs.ctypes.data_as(u64p)))
        return ptr.value, offs

    def buffer_alloc(self, nbytes):
        p = C.c_void_p()
        _chk(lib().dnagpu_buffer_alloc(self.h, nbytes, C.byref(p)))
        return p.value

    def buffer_free(self, ptr):
        lib().dnagpu_buffer_free(self.h, ptr)

    def download_u64(self, ptr, n):
        out = np.empty(max(n, 1), dtype=np.uint64)
        _chk(lib().dnagpu_buffer_download(self.h, ptr, n * 8, out.ctypes.data))
        return out[:n]

    def upload_u64(self, ptr, arr):
        a = np.ascontiguousarray(arr, dtype=np.uint64)
        _chk(lib().dnagpu_buffer_upload(self.h, ptr, a.ctypes.data, a.size * 8))

    # ---- batched operators
    def kmer_hash(self, keys):
        k = np.ascontiguousarray(keys, dtype=np.uint64)
        out = np.empty(max(k.size, 1), dtype=np.uint32)
        _chk(lib().dnagpu_kmer_hash(self.h, k.ctypes.data, k.size, out.ctypes.data, 0))
        return out[:k.size]

    def kmer_match(self, keys, k, flt):
        a = np.ascontiguousarray(keys, dtype=np.uint64)
        out = np.empty(max(a.size, 1), dtype=np.uint8)
        _chk(lib().dnagpu_kmer_match(self.h, a.ctypes.data, a.size, k, C.byref(flt.c), out.ctypes.data, 0))
        return out[:a.size].astype(bool)


MULTI_AUTO, MULTI_RCCL, MULTI_COPY = 0, 1, 2
MULTI_OPT_PARTS, MULTI_OPT_EMULATE_LINK_GBS, MULTI_OPT_PROBE_OWNER, MULTI_OPT_EXCHANGE_RCCL = 1, 2, 3, 4


class _BorrowedContext(Context):
    """a rank's context owned by a Multi: same methods, never destroyed from here"""

    def __init__(self, handle):
        self.h = C.c_void_p(handle)
        # the same test-harness switch as Context: the rank contexts of a Multi are created in C, so the poison flag
        # is set on them here
        self._base_debug = _env_debug()
        if self._base_debug:
            self.set_debug(0)

    def close(self):
        self.h = None


class Multi:
    """dnagpu_multi: N ranks driven from this one process (dnagpu_count_multi)."""

    def __init__(self, devices, transport=MULTI_AUTO):
        n = len(devices)
        arr = (C.c_int * n)(*devices)
        self.h = C.c_void_p()
        _chk(lib().dnagpu_multi_init(arr, n, transport, C.byref(self.h)))
        self.n = n
        self.ranks = [_BorrowedContext(lib().dnagpu_multi_ctx(self.h, r)) for r in range(n)]

    @property
    def transport(self):
        return lib().dnagpu_multi_transport(self.h).decode()

    @property
    def exchange_transport(self):
        """what the most recent count moved its data between ranks with ("peer-copy", "rccl-sendrecv", ...)"""
        return lib().dnagpu_multi_exchange_transport(self.h).decode()

    @property
    def rccl_ranks(self):
        return int(lib().dnagpu_multi_rccl_ranks(self.h))

    def set_exchange_rccl(self, mode):
        """record exchange of count_unordered: 0 = peer copies, 1 = ncclSend / ncclRecv, 2 = own pieces through RCCL too"""
        _chk(lib().dnagpu_multi_set_option(self.h, MULTI_OPT_EXCHANGE_RCCL, float(mode)))

    def rank_phase_times(self, rank):
        """device phases of rank `rank` in the most recent count_unordered: its record pass, then its owner phase"""
        pt = _PhaseTimes()
        _chk(lib().dnagpu_multi_last_phase_times(self.h, rank, C.byref(pt)))
        return [(pt.names[i].decode(), float(pt.ms[i])) for i in range(pt.n)]

    def set_parts(self, parts):
        """bucket groups per owner of count_unordered's pipelined exchange (1 = no overlap)"""
        _chk(lib().dnagpu_multi_set_option(self.h, MULTI_OPT_PARTS, float(parts)))

    def emulate_link(self, gb_per_s):
        """rehearsal on one device: inbound pieces are held to this rate on the transfer stream (0 = off)"""
        _chk(lib().dnagpu_multi_set_option(self.h, MULTI_OPT_EMULATE_LINK_GBS, float(gb_per_s)))

    def probe_owner(self, owner):
        """rehearsal on one device: only this owner pulls and counts (-1: all), so that last_times() are its own"""
        _chk(lib().dnagpu_multi_set_option(self.h, MULTI_OPT_PROBE_OWNER, float(owner)))

    def last_times(self):
        t = _MultiTimes()
        _chk(lib().dnagpu_multi_last_times(self.h, C.byref(t)))
        return {f: getattr(t, f) for f, _ in _MultiTimes._fields_}

    def upload(self, words, n_bases):
        w = np.ascontiguousarray(words, dtype=np.uint64)
        d = C.c_void_p()
        _chk(lib().dnagpu_multi_dna_upload(self.h, w.ctypes.data_as(u64p), n_bases, C.byref(d)))
        return d

    def synth(self, seed, n_bases, motif_len=0):
        d = C.c_void_p()
        _chk(lib().dnagpu_multi_dna_synth(self.h, seed, n_bases, motif_len, C.byref(d)))
        return d

    def dna_free(self, d):
        lib().dnagpu_multi_dna_free(self.h, d)

    def count(self, mdna, k, first=0, count=None):
        """-> [Hist per rank]; their ascending downloads concatenated in rank order are the global result"""
        n_bases = int(lib().dnagpu_multi_dna_length(mdna))
        total = kmer_count(n_bases, k)
        if count is None:
            count = max(total - first, 0)
        hs = (C.c_void_p * self.n)()
        _chk(lib().dnagpu_count_multi(self.h, mdna, k, first, count, hs))
        return [Hist(self.ranks[r], C.c_void_p(hs[r])) for r in range(self.n)]

    def count_unordered(self, mdna, k, first=0, count=None):
        """-> [Hist per rank]: the same groups, disjoint between ranks, in no key order (long k-mers: record exchange)"""
        n_bases = int(lib().dnagpu_multi_dna_length(mdna))
        total = kmer_count(n_bases, k)
        if count is None:
            count = max(total - first, 0)
        hs = (C.c_void_p * self.n)()
        _chk(lib().dnagpu_count_multi_unordered(self.h, mdna, k, first, count, hs))
        return [Hist(self.ranks[r], C.c_void_p(hs[r])) for r in range(self.n)]

    def close(self):
        if self.h:
            lib().dnagpu_multi_destroy(self.h)
            self.h = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()
