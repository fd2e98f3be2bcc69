#!/usr/bin/env python3
"""Per-kernel timing of the backward kernels on DRCT-L training shapes (T = 8 x 32 x 32 tokens), HIP events around
every launch (the library's own profiler).  python tools/bwd_bench.py [--prec bf16] [--iters 20]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from srad_amd import _lib as L
from srad_amd import ops


def timed(label, fn, iters):
    fn()
    torch.cuda.synchronize()
    L.prof_enable(True)
    L.prof_collect()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    prof = L.prof_collect()
    L.prof_enable(False)
    parts = ", ".join(f"{k} {v['ms'] * 1e3 / v['launches']:.1f} us" for k, v in prof.items() if k != "pack_weight")
    print(f"{label:46s} {parts}")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--prec", default="bf16")
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--T", type=int, default=8192)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    T = a.T
    B = T // 1024
    for (N, K) in [(540, 180), (180, 180), (360, 180), (180, 360), (32, 180), (924, 308), (308, 308), (180, 308), (732, 244), (488, 244)]:
        x = torch.randn(T, K, device=dev)
        dy = torch.randn(T, N, device=dev)
        timed(f"wgrad linear M={T} N={N} Cin={K}", lambda: ops.wgrad(dy, x, N, K, precision=a.prec), a.iters)
    for (Cin, Cout, hw) in [(180, 180, 32), (180, 64, 32), (64, 256, 32), (64, 256, 64), (64, 4, 128), (4, 180, 32)]:
        x = torch.randn(B * hw * hw, Cin, device=dev)
        dy = torch.randn(B * hw * hw, Cout, device=dev)
        timed(f"wgrad conv3x3 {hw}x{hw} Cin={Cin} Cout={Cout}", lambda: ops.wgrad(dy, x, Cout, Cin, ntaps=9, B=B, H=hw, W=hw, precision=a.prec), a.iters)
    for d in (180, 308):
        x = torch.randn(T, d, device=dev)
        dy = torch.randn(T, d, device=dev)
        g = torch.randn(d, device=dev)
        timed(f"layernorm_bwd rows={T} C={d}", lambda: ops.layernorm_bwd(dy, x, g, dres=dy), a.iters)
    for d, heads, shift in [(180, 6, 0), (212, 4, 4), (244, 2, 0), (276, 6, 4), (308, 4, 0)]:
        qkv = torch.randn(T, 3 * d, device=dev)
        dout = torch.randn(T, d, device=dev)
        table = torch.randn(225, heads, device=dev)
        timed(f"attn_bwd d={d} heads={heads} shift={shift}", lambda: ops.window_attention_bwd(qkv, dout, table, B, 32, 32, 8, shift, heads), a.iters)
    w = torch.randn(360, 180, device=dev)
    dy = torch.randn(T, 360, device=dev)
    timed("dgrad linear N=360 -> Cin=180", lambda: ops.dgrad(dy, w, precision=a.prec), a.iters)


if __name__ == "__main__":
    main()
