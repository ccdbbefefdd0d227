"""dna-sequences-pg-extension_amd -- MI355X (gfx950) back-end for the k-mer path of the PostgreSQL
extension sid2364/dna-sequences-pg-extension.

The product is the C-ABI shared library ``libdnagpu.so`` (include/dnagpu.h; HIP kernels under
csrc/) plus ``libdna_glue.so`` (glue/, the host-side mirror of the reference's operator surface).
This Python package is only a thin ctypes binding over both, used by the tests, by bench.py and by
the multi-GPU launcher (sharded.py, torch.distributed).  There is no CPU implementation here: if
the library or a GPU is missing every call raises.

The directory name has a hyphen (it mirrors the reference's name), so it is loaded through
``__graft_entry__.load_package()`` under the module name ``dna_sequences_pg_extension_amd``.
"""
from .binding import (  # noqa: F401
    DEBUG_FORCE_SUPERKMER,
    DEBUG_GUARD_POOL,
    DEBUG_HEAVY_EXPAND,
    DEBUG_NO_SLAB0,
    DEBUG_NO_SPEC1,
    DEBUG_SLAB0,
    DEBUG_SLAB0_OVERFLOW,
    DEBUG_SAMPLE1,
    DEBUG_SPEC1_OVERFLOW,
    Context,
    Dna,
    DnaGpuError,
    Filter,
    Hist,
    Records,
    MULTI_AUTO,
    MULTI_COPY,
    MULTI_RCCL,
    Multi,
    abi_version,
    device_count,
    kmer_count,
    lib,
    lib_path,
    strerror,
)
