#!/usr/bin/env python3
"""Loss curves of the DRCT-L x4 training step on a fixed synthetic set (textured 128 px HR images, LR = 4x4 box
average), bf16 mode next to the fp32 (parity) mode with the same seeds - evidence that the bf16 path (bf16 MFMA
operands, fp32 accumulation / master weights / gradients, no loss scaling) trains like the fp32 one.
python tools/train_curve.py [--steps 150] [--batch 8]"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from srad_amd.nets import DRCT
from srad_amd.train import FusedAdam, train_step


class Opt:
    n_colors, img_size, window_size, upscale = 1, 32, 8, 4
    embed_dim, depths, num_heads, mlp_ratio, img_range = 180, (6,) * 12, (6,) * 12, 2, 1.0
    upsampler, resi_connection = "pixelshuffle", "1conv"
    precision, use_graph = "bf16", False


def data(n, seed=0):
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:128, 0:128].astype(np.float32)
    hr = []
    for i in range(n):
        p1, p2 = 5 + rng.uniform(0, 6), 5 + rng.uniform(0, 6)
        t = 127 + 70 * np.sin(2 * np.pi * xx / p1 + rng.uniform(0, 6)) * np.sin(2 * np.pi * yy / p2 + rng.uniform(0, 6))
        hr.append(np.clip(t + rng.normal(0, 3, t.shape), 0, 255))
    hr = np.stack(hr).astype(np.float32)[:, None]
    lr = hr.reshape(n, 1, 32, 4, 32, 4).mean((3, 5))
    return torch.from_numpy(lr).cuda(), torch.from_numpy(hr).cuda()


def run(prec, steps, batch, lr_img, hr_img):
    o = Opt()
    o.precision = prec
    torch.manual_seed(1)
    m = DRCT(o).cuda().train()
    m.enable_training()
    opt = FusedAdam(m, lr=1e-4)
    torch.manual_seed(2)                      # DropPath draws
    losses = []
    t0 = time.perf_counter()
    for it in range(steps):
        i = (it * batch) % (lr_img.shape[0] - batch + 1)
        losses.append(train_step(m, lr_img[i:i + batch], hr_img[i:i + batch], opt))
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    return [float(v) for v in losses], dt


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=150)
    ap.add_argument("--batch", type=int, default=8)
    a = ap.parse_args()
    lr_img, hr_img = data(64)
    out = {}
    curves = {}
    for prec in ("fp32", "bf16"):
        ls, dt = run(prec, a.steps, a.batch, lr_img, hr_img)
        curves[prec] = ls
        out[prec] = {"loss_first": round(ls[0], 3), "loss_at": {str(k): round(ls[k], 3) for k in (10, 50, 100, a.steps - 1) if k < a.steps},
                     "ms_per_step": round(dt / a.steps * 1e3, 2), "finite": bool(np.isfinite(ls).all())}
        out[prec + "_curve"] = [round(v, 3) for v in ls[::10]]
    half = a.steps // 2
    out["max_rel_gap_second_half"] = round(max(abs(b - f) / f for f, b in zip(curves["fp32"][half:], curves["bf16"][half:])), 4)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
