#!/usr/bin/env python3
"""DRN-L x4 training step (SR net + two dual regression models, composite loss) at the C3 shape: RGB, LR [B,3,64,64] ->
256 px HR.  python tools/drn_train_bench.py [--batch 8] [--dtype bf16]"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from srad_amd import _lib as L
from srad_amd.nets import DRN, DownBlock
from srad_amd.train import FusedAdam, drn_train_step


class Opt:
    n_colors, n_blocks, n_feats, negval, rgb_range, scale = 3, 40, 20, 0.2, 255.0, [2, 4]
    precision, use_graph = "bf16", False


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--graph", action="store_true", help="time the step as one replayed hipGraph (train.GraphedDrnTrainStep)")
    ap.add_argument("--tensor-adam", action="store_true", help="the engine's Adam kernel for the dual models (what the Trainer uses) instead of torch.optim.Adam")
    a = ap.parse_args()
    o = Opt()
    o.precision = a.dtype
    torch.manual_seed(1)
    m = DRN(o).cuda().train()
    m.enable_training()
    duals = [DownBlock(o).cuda() for _ in o.scale]
    opt = FusedAdam(m, lr=1e-4, weight_decay=1e-8)
    dopts = [torch.optim.Adam(d.parameters(), lr=1e-4, weight_decay=1e-8) for d in duals]
    if a.tensor_adam:
        from srad_amd.train import TensorAdam
        dopts = [TensorAdam(d.parameters(), lr=1e-4, weight_decay=1e-8) for d in duals]
    B = a.batch
    lrs = [torch.rand(B, 3, 64, 64, device="cuda") * 255, torch.rand(B, 3, 128, 128, device="cuda") * 255]
    hr = torch.rand(B, 3, 256, 256, device="cuda") * 255
    for _ in range(2):
        drn_train_step(m, duals, lrs, hr, opt, dopts)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        loss = drn_train_step(m, duals, lrs, hr, opt, dopts)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / a.steps
    if a.graph:
        from srad_amd.train import GraphedDrnTrainStep, TensorAdam
        gopts = [TensorAdam(d.parameters(), lr=1e-4, weight_decay=1e-8) for d in duals]
        gstep = GraphedDrnTrainStep(m, duals, opt, gopts, warmup=2)
        for _ in range(4):
            gstep(lrs, hr)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(a.steps):
            loss = gstep(lrs, hr)
        torch.cuda.synchronize()
        print(json.dumps({"graphed_ms_per_step": round((time.perf_counter() - t0) / a.steps * 1e3, 2), "eager_ms_per_step": round(dt * 1e3, 2),
                          "loss": float(loss)}))
        return
    L.prof_enable(True)
    drn_train_step(m, duals, lrs, hr, opt, dopts)
    torch.cuda.synchronize()
    L.prof_collect()
    drn_train_step(m, duals, lrs, hr, opt, dopts)
    torch.cuda.synchronize()
    prof = L.prof_collect()
    L.prof_enable(False)
    fl = 3.0 * m.flops(B, 64, 64)
    print(json.dumps({"workload": f"DRN-L x4 train step, RGB 256 px HR, batch {B}", "ms_per_step": round(dt * 1e3, 2),
                      "images_per_s": round(B / dt, 1), "model_tflops": round(fl / dt / 1e12, 1), "loss": float(loss),
                      "kernels": {k: {"launches": v["launches"], "ms": round(v["ms"], 2), "avg_us": round(v["ms"] * 1e3 / v["launches"], 1)}
                                  for k, v in sorted(prof.items(), key=lambda kv: -kv[1]["ms"])}}))


if __name__ == "__main__":
    main()
