#!/usr/bin/env python3
"""BASELINE config C3: DRN-L x4 forward, carpet-shaped RGB input, 256 px HR, batch 8 (LR [8,3,64,64]); time per
batch and per-kernel-class breakdown.  python tools/drn_bench.py [--dtype bf16] [--train]"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from srad_amd import _lib as L
from srad_amd.nets import DRN


class Opt:
    n_colors, n_blocks, n_feats, negval, rgb_range, scale = 3, 40, 20, 0.2, 255.0, [2, 4]
    precision, use_graph = "bf16", True


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--batch", type=int, default=8)
    a = ap.parse_args()
    o = Opt()
    o.precision = a.dtype
    torch.manual_seed(1)
    m = DRN(o).cuda().eval()
    B = a.batch
    x = torch.rand(B, 3, 64, 64, device="cuda") * 255
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
    fl = m.flops(B, 64, 64)
    out = {"workload": f"C3: DRN-L x4 forward, RGB, 256 px HR, batch {B}", "dtype": a.dtype, "ms_per_batch": round(dt * 1e3, 3),
           "hr_mpixels_per_s": round(B * 256 * 256 / dt / 1e6, 2), "gflop": round(fl / 1e9, 1), "model_tflops": round(fl / dt / 1e12, 1),
           "kernels": {k: {"launches": v["launches"], "ms": round(v["ms"], 3), "avg_us": round(v["ms"] * 1e3 / v["launches"], 1),
                           "tflops": round(v["flops"] / (v["ms"] * 1e-3) / 1e12, 1)}
                       for k, v in sorted(prof.items(), key=lambda kv: -kv[1]["ms"])}}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
