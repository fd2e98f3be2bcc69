#!/usr/bin/env python3
"""BASELINE config C5 (DRCT-L eval, one 1024 px HR tile = LR [1,1,256,256], window 64): forward time and the
per-kernel-class breakdown from the library's event profiler.  python tools/c5_bench.py [--dtype bf16]"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from srad_amd import _lib as L
from srad_amd.nets import DRCT


class Opt:
    n_colors, img_size, window_size, upscale = 1, 256, 64, 4
    embed_dim, depths, num_heads, mlp_ratio, img_range = 180, (6,) * 12, (6,) * 12, 2, 1.0
    upsampler, resi_connection = "pixelshuffle", "1conv"
    precision, use_graph = "bf16", True


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--steps", type=int, default=10)
    a = ap.parse_args()
    o = Opt()
    o.precision = a.dtype
    torch.manual_seed(1)
    m = DRCT(o).cuda().eval()
    x = torch.rand(1, 1, 256, 256, device="cuda") * 255
    with torch.no_grad():
        for _ in range(3):
            m(x)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(a.steps):
            m(x)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / a.steps
        m.use_graph = False
        L.prof_enable(True)
        m(x)
        torch.cuda.synchronize()
        L.prof_collect()
        m(x)
        torch.cuda.synchronize()
        prof = L.prof_collect()
        L.prof_enable(False)
    fl = m.flops(1, 256, 256)
    out = {"workload": "C5: DRCT-L x4, LR [1,1,256,256], window 64", "dtype": a.dtype, "ms_per_tile": round(dt * 1e3, 2),
           "hr_mpixels_per_s": round(1024 * 1024 / dt / 1e6, 2), "gflop": round(fl / 1e9, 1), "model_tflops": round(fl / dt / 1e12, 1),
           "kernels": {k: {"launches": v["launches"], "ms": round(v["ms"], 3), "tflops": round(v["flops"] / (v["ms"] * 1e-3) / 1e12, 1)}
                       for k, v in sorted(prof.items(), key=lambda kv: -kv[1]["ms"])}}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
