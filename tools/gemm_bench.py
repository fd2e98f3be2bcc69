#!/usr/bin/env python3
"""Micro-benchmark of the single kernels at the shapes the DRCT-L C2 forward launches them with
(M = 4096 tokens): device microseconds per launch, back-to-back launches timed with HIP events
inside the library (srad_bench_*), so neither Python nor the weight packer is in the number."""
import ctypes as C
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from srad_amd import _lib as L  # noqa: E402

dev = torch.device("cuda:0")
prec = sys.argv[1] if len(sys.argv) > 1 else "bf16"
M = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
P = L.PRECISIONS[prec]


def gemm(K, N, ln=False, res=False, act=0, ldx=None, ntaps=1, B=1, H=1, W=None, hsplit=(0, 0), iters=300):
    W = W or M
    rows = B * H * W
    x = torch.randn(rows, ldx or K, device=dev)
    w = (torch.randn(N, K, ntaps, device=dev) / math.sqrt(K * ntaps)).contiguous()
    b = torch.randn(N, device=dev)
    g, be = (torch.randn(K, device=dev), torch.randn(K, device=dev)) if ln else (None, None)
    r = torch.randn(rows, N, device=dev) if res else None
    ncols = N if not hsplit[0] else (N // hsplit[0]) * hsplit[1]
    y = torch.empty(rows, ncols, device=dev)
    nb = L.lib().srad_op_gemm_scratch_bytes(P, N, K, ntaps)
    scratch = torch.empty(nb + 256, dtype=torch.uint8, device=dev)
    off = (-scratch.data_ptr()) % 256
    us = C.c_float()
    L.check(L.lib().srad_bench_gemm(P, L.dptr(x), x.stride(0), B, H, W, K, L.dptr(w), N, ntaps, 1, L.dptr(b), L.dptr(g),
                                    L.dptr(be), act, L.dptr(r), N if res else 0, L.dptr(y), ncols, hsplit[0], hsplit[1],
                                    C.c_void_p(scratch.data_ptr() + off), C.c_size_t(nb), iters, C.byref(us),
                                    L.current_stream_ptr()), "bench_gemm")
    return us.value


def attn(d, heads, shift, iters=300):
    hd = d // heads
    hdp = (hd + 3) // 4 * 4
    B = M // 1024
    qkv = torch.randn(M, 3 * heads * hdp, device=dev)
    table = torch.randn(225, heads, device=dev)
    out = torch.empty(M, d, device=dev)
    us = C.c_float()
    L.check(L.lib().srad_bench_window_attn(P, L.dptr(qkv), L.dptr(out), L.dptr(table), B, 32, 32, 8, shift, d, heads, hdp,
                                           iters, C.byref(us), L.current_stream_ptr()), "bench_attn")
    return us.value


def main():
    print(f"precision {prec}, M = {M}")
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        tot = 0.0
        for name, K, N, kw, mult in [
            ("qkv d180 (LN)", 180, 540, dict(ln=True, ldx=308, hsplit=(30, 32)), 12),
            ("qkv d212 (LN)", 212, 636, dict(ln=True, ldx=308, hsplit=(53, 56)), 12),
            ("qkv d244 (LN)", 244, 732, dict(ln=True, ldx=308, hsplit=(122, 124)), 12),
            ("qkv d276 (LN)", 276, 828, dict(ln=True, ldx=308, hsplit=(46, 48)), 12),
            ("qkv d308 (LN)", 308, 924, dict(ln=True, ldx=308, hsplit=(77, 80)), 12),
            ("proj d180 (+res)", 180, 180, dict(res=True), 12),
            ("proj d308 (+res)", 308, 308, dict(res=True), 12),
            ("fc1 d180 (LN+gelu)", 180, 360, dict(ln=True, act=1), 12),
            ("fc1 d244 (LN+gelu)", 244, 488, dict(ln=True, act=1), 12),
            ("fc2 d180 (+res)", 360, 180, dict(res=True), 12),
            ("fc2 d244 (+res)", 488, 244, dict(res=True), 12),
            ("adjust d180", 180, 32, dict(act=2), 48),
            ("adjust5 d308", 308, 180, dict(res=True), 12),
            ("conv3x3 180->180", 180, 180, dict(ntaps=9, B=M // 1024, H=32, W=32, res=True), 1),
            ("conv3x3 64->256", 64, 256, dict(ntaps=9, B=M // 1024, H=32, W=32), 1),
        ]:
            t = gemm(K, N, **kw)
            fl = 2.0 * M * K * N * kw.get("ntaps", 1)
            tot += t * mult
            print(f"{name:22s} K={K:4d} N={N:4d}: {t:8.2f} us/launch  ({fl / t / 1e6:8.1f} TFLOP/s)")
        for d, heads, shift in [(180, 6, 0), (212, 4, 4), (244, 2, 0), (276, 6, 4), (308, 4, 0)]:
            t = attn(d, heads, shift)
            tot += 12 * t
            print(f"attn d={d} h={heads} shift={shift}: {t:8.2f} us/launch")
    torch.cuda.synchronize()


if __name__ == "__main__":
    main()
