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
attn = torch.randn(M, 320, device=dev)
short = torch.randn(M, 320, device=dev)
y = torch.empty(M, 320, device=dev)
w = torch.randn(512 * 512, device=dev) * 0.05
scratch = torch.zeros(8 << 20, dtype=torch.uint8, device=dev)
off = (-scratch.data_ptr()) % 256
align = lambda v: (v + 255) // 256 * 256
kp = lambda n, k: align(((n + 63) // 64 * 64) * ((k + 31) // 32 * 32) * 2)       # srad_packed_bytes(bf16, n, k, 1), 256-aligned
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    for d, m, no in [(180, 360, 32), (308, 308, 180)]:
        us = C.c_float()
        L.check(L.lib().srad_bench_mlp_block(M, d, m, no, L.dptr(attn), L.dptr(short), L.dptr(y), L.dptr(w),
                                             C.c_void_p(scratch.data_ptr() + off), C.c_size_t(scratch.numel() - off), 0x10000, 20,
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
