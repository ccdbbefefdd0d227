"""Worker of test_pool_buffer_on_another_stream: a pooled device buffer (dnagpu_partition_kmers' keys) consumed by HIP
work on a NON-default torch stream, following the stream rule of include/dnagpu.h: the caller's stream is ordered
behind dnagpu_stream() before it touches the buffer (here: ctx.synchronize()), and is finished before the buffer goes
back to the pool.  The pool then recycles the block for the next call; both calls must give the right sums."""
import ctypes as C
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import oracle as orc  # noqa: E402
from __graft_entry__ import load_package  # noqa: E402


class _DevArray:
    def __init__(self, ptr, n):
        self.__cuda_array_interface__ = {"shape": (n,), "typestr": "<i8", "data": (int(ptr), False), "version": 2, "strides": None}


def main():
    out_path = sys.argv[1]
    pkg = load_package()
    dev = torch.device("cuda", 0)
    side = torch.cuda.Stream(device=dev)
    ok = True
    detail = []
    with pkg.Context(0) as ctx:
        for rnd, (n, k) in enumerate(((400_000, 31), (400_000, 31), (300_000, 27))):
            d = ctx.synth(0xD2A0003 + rnd, n)
            words = d.download()
            keys = orc.generate_kmers(words, n, k, faithful=False)
            ptr, offs = ctx.partition_kmers(d, k, 0, len(keys), 4)
            ctx.synchronize()                      # rule (1): the caller's stream starts behind dnagpu_stream()
            with torch.cuda.stream(side):
                t = torch.as_tensor(_DevArray(ptr, len(keys)), device=dev)
                s = [int((t[int(offs[o]):int(offs[o + 1])] & 0xFFFF).sum().item()) for o in range(4)]
                total = int((t & 0xFFFF).sum().item())
            side.synchronize()                     # rule (2): finished before the buffer goes back to the pool
            ctx.buffer_free(ptr)
            want = int((keys & np.uint64(0xFFFF)).sum())
            good = total == want and sum(s) == want
            ok = ok and good
            detail.append({"round": rnd, "total": total, "want": want})
            d.free()
    with open(out_path, "w") as f:
        json.dump({"ok": ok, "detail": detail}, f)


if __name__ == "__main__":
    main()
