"""CPU-side checks of the drop-in boundary: libdnagpu.so loads and exports every symbol that
include/dnagpu.h declares; without a GPU the product fails loudly (no CPU fallback)."""
import ctypes as C
import os
import re

import pytest

from __graft_entry__ import ROOT, load_package


def declared_symbols(header):
    text = open(header).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(dnagpu_[a-z0-9_]+)\s*\(", text)))


def test_exports_every_declared_symbol():
    pkg = load_package()
    names = declared_symbols(os.path.join(ROOT, "include", "dnagpu.h"))
    assert len(names) >= 30
    L = C.CDLL(pkg.lib_path())
    missing = [n for n in names if not hasattr(L, n)]
    assert not missing, f"declared in dnagpu.h but not exported: {missing}"


def test_error_texts_are_the_references():
    pkg = load_package()
    assert pkg.strerror(1) == "Invalid k value: must be between 1 and 32"            # dna.c:773
    assert pkg.strerror(2) == "Qkmer pattern and kmer lengths do not match"          # dna.c:1107
    assert pkg.strerror(3) == "Prefix length cannot exceed kmer length"              # dna.c:855
    assert pkg.abi_version() == 2


def test_kmer_count_rule():
    pkg = load_package()
    assert pkg.kmer_count(19, 5) == 15            # test.sql:95: 15 rows
    assert pkg.kmer_count(7, 7) == 1
    assert pkg.kmer_count(7, 8) == 0
    assert pkg.kmer_count(2, 32) == 0             # the reference underflows here (dna.c:781)
    for k in (0, 33, -5):
        with pytest.raises(pkg.DnaGpuError) as ei:
            pkg.kmer_count(100, k)
        assert ei.value.code == 1


def test_no_cpu_fallback_without_gpu():
    pkg = load_package()
    try:
        import torch
        has_gpu = torch.cuda.is_available()
    except Exception:
        has_gpu = os.path.exists("/dev/kfd")
    if has_gpu:
        pytest.skip("a GPU is present")
    with pytest.raises(pkg.DnaGpuError) as ei:
        pkg.Context(0)
    assert ei.value.code == 7                     # DNAGPU_ERR_NO_DEVICE: fails loudly, no fallback


def test_key_mix_is_a_bijection_of_the_key_bits(tmp_path):
    """key_mix / key_unmix (kmer_device.hpp; the keys of the oversize buckets travel mixed through their tree and the groups
    are turned back): a bijection of the 2k-bit keys for every k the engine takes, near-copies spread over the top digits.
    Host code of the same header, built with hipcc: no device is touched."""
    import shutil
    import subprocess
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "keymix_check")
    subprocess.check_call([hipcc, "-O1", "-std=c++17", "--offload-arch=gfx950", "-I",
                           os.path.join(root, "dna-sequences-pg-extension_amd", "csrc"),
                           os.path.join(root, "tests", "host", "keymix_check.cpp"), "-o", exe])
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and r.stdout.strip().endswith("ok"), r.stdout + r.stderr
