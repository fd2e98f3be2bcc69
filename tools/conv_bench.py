#!/usr/bin/env python3
"""DRN conv shapes through the row-gather GEMM: microseconds per launch (srad_bench_gemm).  python tools/conv_bench.py [bf16]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.argv = [sys.argv[0]] + (sys.argv[1:2] or ["bf16"]) + ["8192"]
import gemm_bench as G  # noqa: E402
import torch  # noqa: E402

s = torch.cuda.Stream()
with torch.cuda.stream(s):
    for name, K, N, B, H, W, taps in [("rcab 80->80 @64x64 b8", 80, 80, 8, 64, 64, 9), ("rcab 80->80 @128x128 b8", 80, 80, 8, 128, 128, 9),
                                      ("1x1 80->80 @128x128 b8", 80, 80, 8, 128, 128, 1), ("1x1 720->80 @128x128 b8", 720, 80, 8, 128, 128, 1),
                                      ("up 80->320 @64x64 b8", 80, 320, 8, 64, 64, 9), ("rcab 80->80 @128x128 b1", 80, 80, 1, 128, 128, 9)]:
        G.M = B * H * W
        t = G.gemm(K, N, ntaps=taps, B=B, H=H, W=W, iters=20)
        fl = 2.0 * B * H * W * K * N * taps
        print(f"{name:28s}: {t:8.1f} us  {fl / t / 1e6:7.1f} TFLOP/s")
