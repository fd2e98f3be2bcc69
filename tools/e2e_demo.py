#!/usr/bin/env python3
"""End-to-end run of the reference's workflow on synthetic MVTec-grid-shaped data, all on the HIP engine:
train DRCT-L x4 (bf16) on defect-free textures -> super-resolve a test split of 21 good + 57 defective tiles ->
SSIM window sweep / MSE / PSNR -> the three AUCs; then the SAME trained weights through (a) the engine in fp32 mode and in the
split-bf16 mode (the evaluator's default) and (b) the CPU oracle (fp32 torch + numpy scorer), to check the north-star bar |dAUC| <= 0.002 in a regime where the
AUC is not at chance.  python tools/e2e_demo.py [--steps 300] [--rdg 12]"""
import argparse
import contextlib
import io
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from srad_amd import evaluate as E
from srad_amd.nets import DRCT
from srad_amd.train import FusedAdam, train_step


def textures(n, rng, defect=False):
    yy, xx = np.mgrid[0:128, 0:128].astype(np.float32)
    out = []
    for _ in range(n):
        p1, p2 = 14 + rng.uniform(0, 8), 14 + rng.uniform(0, 8)     # well below the LR Nyquist limit: learnable
        t = 127 + 70 * np.sin(2 * np.pi * xx / p1 + rng.uniform(0, 6)) * np.sin(2 * np.pi * yy / p2 + rng.uniform(0, 6))
        t = t + rng.normal(0, 1, t.shape)
        if defect:                                  # a small patch of extra pixel noise: detail the 4x4 averaging destroys and SR cannot invent
            cy, cx = rng.integers(24, 104, 2)
            r = rng.uniform(3.0, 9.0)
            t = np.where((yy - cy) ** 2 + (xx - cx) ** 2 < r * r, t + rng.normal(0, rng.uniform(4, 30), t.shape), t)
        out.append(np.clip(t, 0, 255))
    return np.stack(out).astype(np.float32)


def lr_of(hr):
    return hr.reshape(hr.shape[0], 32, 4, 32, 4).mean((2, 4))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--rdg", type=int, default=12)
    a = ap.parse_args()

    class Opt:
        n_colors, img_size, window_size, upscale = 1, 32, 8, 4
        embed_dim, depths, num_heads, mlp_ratio, img_range = 180, (6,) * a.rdg, (6,) * a.rdg, 2, 1.0
        upsampler, resi_connection = "pixelshuffle", "1conv"
        precision, use_graph, rgb_range = "bf16", False, 255.0

    rng = np.random.default_rng(0)
    train_hr = textures(128, rng)
    tl, th = torch.from_numpy(lr_of(train_hr))[:, None].cuda(), torch.from_numpy(train_hr)[:, None].cuda()
    torch.manual_seed(1)
    m = DRCT(Opt()).cuda().train()
    m.enable_training()
    opt = FusedAdam(m, lr=2e-4)
    t0 = time.perf_counter()
    losses = []
    for it in range(a.steps):
        i = (it * 8) % 121
        losses.append(train_step(m, tl[i:i + 8], th[i:i + 8], opt))
    torch.cuda.synchronize()
    train_s = time.perf_counter() - t0
    losses = [float(v) for v in losses]

    good_hr, bad_hr = textures(21, rng), textures(57, rng, defect=True)

    def pairs(hr):
        u8 = np.clip(np.rint(hr), 0, 255).astype(np.uint8)
        lr = np.clip(np.rint(lr_of(hr)), 0, 255).astype(np.uint8)
        return [(lr[i][:, :, None], u8[i][:, :, None]) for i in range(hr.shape[0])]
    good, bad = pairs(good_hr), pairs(bad_hr)
    res = {}
    with contextlib.redirect_stdout(io.StringIO()):
        res["hip_bf16"] = E.evaluate_on_test(Opt, m, good, bad)
        o32 = Opt()
        o32.precision = "fp32"
        m32 = DRCT(o32).cuda().eval()
        m32.load_state_dict(m.state_dict())
        res["hip_fp32"] = E.evaluate_on_test(Opt, m32, good, bad)
        o3 = Opt()
        o3.precision = "bf16x3"                                  # split-bf16: the evaluator's default since round 3
        m3 = DRCT(o3).cuda().eval()
        m3.load_state_dict(m.state_dict())
        res["hip_bf16x3"] = E.evaluate_on_test(Opt, m3, good, bad)
    # CPU oracle with the trained weights
    from oracle import scorer_ref as O
    from oracle import sr_ref as R
    sd = {k: v.detach().cpu() for k, v in m.state_dict().items()}
    allp = good + bad
    sr = []
    with torch.no_grad():
        for i in range(0, len(allp), 6):
            x = torch.from_numpy(np.stack([p[0] for p in allp[i:i + 6]])).permute(0, 3, 1, 2).float()
            sr += [np.transpose(O.to_u8_trunc(t), (1, 2, 0)) for t in R.drct_forward(sd, x, m.cfg).numpy()]
    res["oracle_cpu"] = O.evaluate_pairs([0] * len(good) + [1] * len(bad), sr, [p[1] for p in allp])
    keys = ("auc_ssim", "auc_mse", "auc_psnr")
    out = {"train": {"steps": a.steps, "seconds": round(train_s, 1), "loss_first": round(losses[0], 2), "loss_last": round(losses[-1], 2)},
           "auc": {k: {q: round(float(v[q]), 4) for q in keys} | {"best_ws": int(v["best_ws"])} for k, v in res.items()},
           "max_abs_auc_diff_fp32_mode_vs_oracle": round(max(abs(res["hip_fp32"][q] - res["oracle_cpu"][q]) for q in keys), 5),
           "max_abs_auc_diff_bf16_mode_vs_oracle": round(max(abs(res["hip_bf16"][q] - res["oracle_cpu"][q]) for q in keys), 5),
           "max_abs_auc_diff_bf16x3_mode_vs_oracle": round(max(abs(res["hip_bf16x3"][q] - res["oracle_cpu"][q]) for q in keys), 5)}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
