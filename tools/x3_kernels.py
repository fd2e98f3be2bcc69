#!/usr/bin/env python3
"""The two fused Swin-block kernels alone on the chip, bf16 against split-bf16 ("bf16x3"), per block shape of DRCT-L:
microseconds per launch of back-to-back launches (srad_bench_mlp_block / srad_bench_qkv_attn; bit 17 = the split kernel)."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from srad_amd import _lib as L

dev = torch.device("cuda:0")
SPLIT = 0x20000
M = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
B = M // 1024
attn_h = torch.randn(M, 320, device=dev).to(torch.bfloat16)
attn_f = torch.randn(M, 320, device=dev)
short = torch.randn(M, 320, device=dev)
y = torch.empty(M, 320, device=dev)
w = torch.randn(1024 * 512, device=dev) * 0.05      # >= 3 d^2 floats (the qkv weight of d = 308)
scratch = torch.empty(32 << 20, dtype=torch.uint8, device=dev)
off = (-scratch.data_ptr()) % 256
sp, sb = C.c_void_p(scratch.data_ptr() + off), C.c_size_t(scratch.numel() - off)
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    for d, heads, m, no in [(180, 6, 360, 32), (212, 4, 424, 32), (244, 2, 488, 32), (276, 6, 276, 32), (308, 4, 308, 180)]:
        us = C.c_float()
        row = []
        for name, flags, a in (("bf16 fm16", 16 << 8, attn_h), ("x3 fm16", (16 << 8) | SPLIT, attn_f), ("x3 fm32", (32 << 8) | SPLIT, attn_f)):
            L.check(L.lib().srad_bench_mlp_block(M, d, m, no, L.dptr(a), L.dptr(short), L.dptr(y), L.dptr(w), sp, sb, flags, 100, C.byref(us),
                                                 L.current_stream_ptr()), "bench_mlp_block")
            row.append(f"{name} {us.value:6.2f}")
        qrow = []
        for name, flags in (("bf16", 4), ("x3", 4 | SPLIT)):
            L.check(L.lib().srad_bench_qkv_attn(L.dptr(short), 320, B, 32, 32, flags, d, heads, L.dptr(w), L.dptr(y), sp, sb, 100, C.byref(us),
                                                L.current_stream_ptr()), "bench_qkv_attn")
            qrow.append(f"{name} {us.value:6.2f}")
        print(f"M={M} d={d} heads={heads}: mlp_block [" + " | ".join(row) + "]   qkv_attn [" + " | ".join(qrow) + "] us", flush=True)
torch.cuda.synchronize()
