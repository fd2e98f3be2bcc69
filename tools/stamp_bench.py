#!/usr/bin/env python3
"""Where a 16-row mlp_block workgroup spends its cycles: the diagnostic (STAMP) build of the kernel writes s_memtime at every
phase boundary for every wave; this prints the median over workgroups of each segment, for wave 0 and wave 7, in shader
cycles.  Shares only: the stamps' own waits make the build slower than the real kernel."""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from srad_amd import _lib as L  # noqa: E402

NAMES = ["entry", "loads issued", "vectors arrived", "proj: prev done", "proj: barrier", "proj: MFMAs issued", "fc1: epilogue(LN2) done",
         "fc1: barrier", "fc1: MFMAs issued", "fc2: epilogue(GELU) done", "fc2: barrier", "fc2: MFMAs issued", "adj: epilogue done",
         "adj: barrier", "adj: MFMAs issued", "end"]
dev = torch.device("cuda:0")
M = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
attn = torch.randn(M, 320, device=dev).to(torch.bfloat16)
short = torch.randn(M, 320, device=dev)
y = torch.empty(M, 320, device=dev)
w = torch.randn(1024 * 1024, device=dev) * 0.05          # >= 3 d x d floats for the widest block (924 x 308)
scratch = torch.zeros(8 << 20, dtype=torch.uint8, device=dev)
off = (-scratch.data_ptr()) % 256
align = lambda v: (v + 255) // 256 * 256
kp = lambda n, k: align(((n + 63) // 64 * 64) * ((k + 31) // 32 * 32) * 2)       # srad_packed_bytes(bf16, n, k, 1), 256-aligned
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    for d, m, no, dbg in [(180, 360, 32, 0), (180, 360, 32, 1), (308, 308, 180, 0), (308, 308, 180, 1)]:
        us = C.c_float()
        print("switch-off bits:", dbg, "(1 = no weight loads)")
        L.check(L.lib().srad_bench_mlp_block(M, d, m, no, L.dptr(attn), L.dptr(short), L.dptr(y), L.dptr(w),
                                             C.c_void_p(scratch.data_ptr() + off), C.c_size_t(scratch.numel() - off), 0x10000 | dbg, 20,
                                             C.byref(us), L.current_stream_ptr()), "bench")
        torch.cuda.synchronize()
        base = off + kp(d, d) + kp(m, d) + kp(d, m) + kp(no, d)
        st = scratch[base:base + (M // 16) * 8 * 16 * 8].view(torch.int64).view(M // 16, 8, 16).cpu().numpy().astype(np.int64)
        rel = st - st[:, :1, :1]                                           # cycles since wave 0 of the workgroup entered
        print(f"d={d} m={m} no={no}: {us.value:.1f} us per launch (stamp build)")
        for wv in (0, 7):
            med = np.median(rel[:, wv, :], axis=0)
            print(f"  wave {wv}: " + "  ".join(f"{NAMES[i]}={int(med[i])}" for i in range(16)))
            seg = np.diff(med)
            print("     segments (cycles): " + "  ".join(f"{NAMES[i + 1]}:{int(seg[i])}" for i in range(15)))
        start = st[:, 0, 0] - st[:, 0, 0].min()
        end = st[:, :, 15].max(axis=1) - st[:, 0, 0].min()
        print(f"  workgroup start spread: median {int(np.median(start))} max {int(start.max())} cycles; last end {int(end.max())}; "
              f"median workgroup lifetime {int(np.median(st[:, :, 15].max(axis=1) - st[:, 0, 0]))}")

# ---- the first half (qkv_attn): same stamps ----
QN = ["entry", "loads issued", "vectors staged", "x rows arrived + stats", "barrier(gamma)", "xn written", "w look-ahead issued", "barrier(xn)",
      "qkv stages done", "barrier(qkv)", "softmax done", "barrier(P)", "-", "-", "-", "end"]
x = torch.randn(M, 320, device=dev)
out = torch.empty(M, 320, device=dev)
with torch.cuda.stream(s):
    for d, heads in [(180, 6), (244, 2), (276, 6), (308, 4)]:
        us = C.c_float()
        hdt = ((d // heads) + 15) // 16
        wb = align(heads * 3 * 16 * hdt * ((d + 31) // 32 * 32) * 2)
        scratch.zero_()
        L.check(L.lib().srad_bench_qkv_attn(L.dptr(x), 320, 4, 32, 32, 4 | 0x10000, d, heads, L.dptr(w), L.dptr(out),
                                            C.c_void_p(scratch.data_ptr() + off), C.c_size_t(scratch.numel() - off), 20, C.byref(us),
                                            L.current_stream_ptr()), "bench")
        torch.cuda.synchronize()
        nwg = 64 * heads
        st = scratch[off + wb:off + wb + nwg * 8 * 16 * 8].view(torch.int64).view(nwg, 8, 16).cpu().numpy().astype(np.int64)
        rel = st - st[:, :1, :1]
        print(f"qkv_attn d={d} heads={heads}: {us.value:.1f} us per launch (stamp build), {nwg} workgroups")
        for wv in (0, 7):
            med = np.median(rel[:, wv, :], axis=0)
            idx = [0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 15]
            print(f"  wave {wv}: " + "  ".join(f"{QN[i]}={int(med[i])}" for i in idx))
