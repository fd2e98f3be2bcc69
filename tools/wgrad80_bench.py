#!/usr/bin/env python3
"""Stand-alone timing of 3x3 convolution weight gradients at DRN-L training shapes (batch 8, LR 64 px): 80 -> 80
channels at 64 and 128 px (the RCAB convolutions), 40 -> 40 at 128 px, 20 -> 20 at 256 px.  HIP events around every launch (the library's profiler).
python tools/wgrad80_bench.py [--iters 20]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from srad_amd import _lib as L
from srad_amd import ops


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--ksplit", type=int, default=0, help="override the number of row splits (SRAD_WGRAD_KSPLIT)")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    B = a.batch
    if a.ksplit:
        os.environ["SRAD_WGRAD_KSPLIT"] = str(a.ksplit)
    for (C, px) in [(80, 64), (80, 128), (40, 128), (20, 256)]:
        M = B * px * px
        x = torch.randn(M, C, device=dev)
        dy = torch.randn(M, C, device=dev)
        fn = lambda: ops.wgrad(dy, x, C, C, ntaps=9, B=B, H=px, W=px, precision="bf16")
        fn()
        torch.cuda.synchronize()
        L.prof_enable(True)
        L.prof_collect()
        for _ in range(a.iters):
            fn()
        torch.cuda.synchronize()
        prof = L.prof_collect()
        L.prof_enable(False)
        fl = 2.0 * M * C * C * 9
        parts = ", ".join(f"{k} {v['ms'] * 1e3 / v['launches']:.1f} us" for k, v in prof.items() if k != "pack_weight")
        w = prof["wgrad"]["ms"] * 1e3 / prof["wgrad"]["launches"]
        print(f"{C:3d} -> {C:3d} ch, {px:3d} px, M {M:7d}: {parts}   ({fl / w / 1e6:.0f} TFLOP/s wgrad alone)")


if __name__ == "__main__":
    main()
