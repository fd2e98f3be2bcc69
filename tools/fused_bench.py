#!/usr/bin/env python3
"""Timing experiments on the fused MLP-block kernel (srad_bench_mlp_block): which part of a stage costs what."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from srad_amd import _lib as L  # noqa: E402

dev = torch.device("cuda:0")
M = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
attn = torch.randn(M, 320, device=dev).to(torch.bfloat16)
short = torch.randn(M, 320, device=dev)
y = torch.empty(M, 320, device=dev)
w = torch.randn(512 * 512, device=dev) * 0.05
scratch = torch.empty(4 << 20, dtype=torch.uint8, device=dev)
off = (-scratch.data_ptr()) % 256
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    for d, m, no in [(180, 360, 32), (212, 424, 32), (244, 488, 32), (276, 276, 32), (308, 308, 180)]:      # DRCT-L's five Swin blocks
        row = []
        for dbg in (0, 1, 2, 4, 3, 16, 19, 51):
            us = C.c_float()
            L.check(L.lib().srad_bench_mlp_block(M, d, m, no, L.dptr(attn), L.dptr(short), L.dptr(y), L.dptr(w),
                                                 C.c_void_p(scratch.data_ptr() + off), C.c_size_t(scratch.numel() - off), dbg, 100,
                                                 C.byref(us), L.current_stream_ptr()), "bench")
            row.append(f"dbg{dbg}={us.value:6.1f}")
        print(f"d={d} m={m} no={no}: " + "  ".join(row) + "   (15 = no W load/MFMA/GELU/W store; +16 no epilogues; +32 no shortcut loads)")
torch.cuda.synchronize()
