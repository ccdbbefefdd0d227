"""Leaves-kernel time per leaf for crafted leaves: 2048 leaves (leaf id = top 11 key bits, so the tree's two
levels isolate them) of `heavy` copies of one key + `light` random keys each.  Usage: leaf_probe.py"""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import load_package  # noqa: E402

pkg = load_package()
L = 2048
rng = np.random.default_rng(5)
with pkg.Context(0) as ctx:
    ctx.set_profiling(True)
    for heavy, light in ((0, 5477), (5000, 477), (5000, 0), (2500, 2977), (1000, 4477), (200, 5277), (5000, 1000)):
        per = heavy + light
        ids = np.repeat(np.arange(L, dtype=np.uint64), per) << np.uint64(51)
        pay = rng.integers(0, 2**51, L * per, dtype=np.uint64)
        if heavy:
            pay.reshape(L, per)[:, :heavy] = np.uint64(0x1234567890ABC)
        keys = ids | pay
        rng.shuffle(keys)
        t = torch.from_numpy(keys.view(np.int64)).cuda()
        best = {}
        for _ in range(3):
            h = ctx.count_keys_device(C.c_void_p(t.data_ptr()), t.numel(), 31)
            for a, b in ctx.last_phase_times():
                best[a] = min(best.get(a, 1e9), b)
            dist = h.distinct
            h.free()
        lv = best.get("leaves", 0.0)
        print(f"heavy {heavy:5d} light {light:5d}: leaves {lv:7.3f} ms = {lv * 1e3 / (L / 512):7.1f} us per leaf per workgroup, "
              f"groups {dist}, phases {[(a, round(b, 2)) for a, b in best.items() if b > 0.05]}", flush=True)
