"""The CPU oracle under AddressSanitizer + UBSan (GPU sanitizers are not available on the pool, so
memory checking happens on the CPU build).  Runs a spread of oracle calls in a child process with
libasan preloaded."""
import os
import subprocess
import sys

import pytest

from __graft_entry__ import ROOT

SCRIPT = r"""
import ctypes as C, numpy as np, os, sys
sys.path.insert(0, %r)
import oracle as orc
orc._LIB = None
so = os.path.join(%r, "oracle", "liboracle_asan.so")
orc.build = lambda force=False: so
L = orc.lib()
for n in (1, 31, 32, 33, 64, 65, 1000, 4097):
    w = orc.synth_words(7 + n, n)
    for k in (1, 5, 21, 31, 32):
        a = orc.generate_kmers(w, n, k, faithful=True)
        b = orc.generate_kmers(w, n, k, faithful=False)
        assert np.array_equal(a, b)
        if n >= k:
            keys, counts = orc.count_kmers(w, n, k)
            assert int(counts.sum()) == n - k + 1
    orc.generate_kmers_contains(w, n, 1, "N") if n >= 1 else None
w, n = orc.dna_encode("ATCGATCGATCGATCGACG")
assert orc.hist_summary(*orc.count_kmers(w, n, 5))[:3] == (15, 6, 2)
w = orc.synth_words_repeat(3, 5000, 17)
orc.count_kmers(w, 5000, 32)
print("asan-ok")
"""


def test_oracle_under_asan_ubsan():
    libasan = subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    if not libasan or not os.path.exists(libasan):
        pytest.skip("libasan not available")
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "liboracle_asan.so"])
    env = dict(os.environ, LD_PRELOAD=libasan, ASAN_OPTIONS="detect_leaks=0:halt_on_error=1",
               UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1")
    r = subprocess.run([sys.executable, "-c", SCRIPT % (ROOT, ROOT)], env=env, capture_output=True, text=True,
                       timeout=300)
    assert r.returncode == 0 and "asan-ok" in r.stdout, r.stdout + r.stderr
