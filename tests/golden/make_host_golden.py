#!/usr/bin/env python3
"""Generate tests/golden/host_golden.npz by running THE REFERENCE's own loss types (src/loss.py) and sample-preparation
functions (src/data.py: get_patch, augment, set_channel, np2Tensor) on deterministic synthetic inputs.  Data only:
inputs, seeds and expected outputs.  Build container only (/root/reference does not exist on the GPU box):

    python tests/golden/make_host_golden.py

Same import shim as make_golden.py: the reference imports skimage / imageio / matplotlib at module top; the first two
are absent from the image and unused by the functions called here (empty stand-in modules), matplotlib is only used
by the plot methods that are never called."""
import os
import random
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = os.environ.get("SRAD_REFERENCE", "/root/reference")
sys.dont_write_bytecode = True
sys.path.insert(0, ROOT)


def _stub(names):
    for n in names:
        if n not in sys.modules:
            try:
                __import__(n)
            except Exception:
                sys.modules[n] = types.ModuleType(n)
    for n in names:
        if "." in n:
            parent, child = n.rsplit(".", 1)
            setattr(sys.modules[parent], child, sys.modules[n])


def main():
    _stub(["skimage", "skimage.color", "imageio", "imageio.v2", "matplotlib", "matplotlib.pyplot"])
    sys.path.insert(0, REF)
    import torch
    from src import loss as RL
    from src import data as RD
    torch.manual_seed(0)
    out = {}

    class A:
        rgb_range, batch_size = 255, 3

    def images(tag, B, C, H, W, extra=0):
        import zlib
        g = torch.Generator().manual_seed(zlib.crc32(tag.encode()) % 1000 + 7)
        yy, xx = torch.meshgrid(torch.arange(H + extra, dtype=torch.float32), torch.arange(W + extra, dtype=torch.float32), indexing="ij")
        base = 127 + 90 * torch.sin(xx / 5.0) * torch.cos(yy / 7.0)
        hr = (base[None, None, :H, :W] + 25 * torch.randn(B, C, H, W, generator=g)).clamp(-20, 280)       # some values outside [0, 255]: the clamp
        sr = (base[None, None] + 25 * torch.randn(B, C, H + extra, W + extra, generator=g) + 6).clamp(-20, 280)
        return sr, hr

    cases = {"gray48": (2, 1, 48, 40, 0), "rgb40": (2, 3, 40, 44, 0), "gray16": (1, 1, 16, 16, 0), "gray_crop": (1, 1, 32, 36, 4)}
    for tag, (B, C, H, W, extra) in cases.items():
        sr, hr = images(tag, B, C, H, W, extra)
        out[f"loss/{tag}/sr"], out[f"loss/{tag}/hr"] = sr.numpy(), hr.numpy()
        specs = ["1*SSIM"] if extra else ["1*L1", "1*MSE", "1*PSNR", "1*SSIM", "0.7*L1+0.3*SSIM", "1*MSE+0.05*PSNR"]
        for spec in specs:
            A.loss = spec
            lf = RL.Loss(A, None)
            lf.start_log()
            x = sr.clone().requires_grad_(True)
            val = lf(x, hr)
            val.backward()
            key = spec.replace("*", "x").replace("+", "_")
            out[f"loss/{tag}/{key}/value"] = np.array(val.item(), dtype=np.float64)
            out[f"loss/{tag}/{key}/grad"] = x.grad.numpy()
            out[f"loss/{tag}/{key}/log"] = lf.log.numpy().copy()
            print(tag, spec, "value %.6f  |grad| %.3e" % (val.item(), x.grad.abs().max()))

    # ---- get_patch / augment under fixed seeds (src/data.py:21-50): index images so every pixel is identifiable ----
    hr = np.arange(64 * 72, dtype=np.int32).reshape(64, 72, 1)
    lr4 = np.arange(16 * 18, dtype=np.int32).reshape(16, 18, 1) + 100000
    lr2 = np.arange(32 * 36, dtype=np.int32).reshape(32, 36, 1) + 200000
    out["data/hr"], out["data/lr4"], out["data/lr2"] = hr, lr4, lr2
    for seed in range(6):
        random.seed(seed)
        (pl, ph) = RD.get_patch([lr4, lr2], hr, patch_size=32, scale=[4, 2], multi_scale=True)
        al, ah = RD.augment(pl, ph)
        out[f"data/seed{seed}/patch_lr4"], out[f"data/seed{seed}/patch_lr2"], out[f"data/seed{seed}/patch_hr"] = pl[0], pl[1], ph
        out[f"data/seed{seed}/aug_lr4"], out[f"data/seed{seed}/aug_lr2"], out[f"data/seed{seed}/aug_hr"] = (
            np.ascontiguousarray(al[0]), np.ascontiguousarray(al[1]), np.ascontiguousarray(ah))
    random.seed(3)
    (pl, ph) = RD.get_patch([lr4], hr[:, :64], patch_size=64, scale=[4])          # patch = whole image (the CLI's case)
    out["data/full/patch_lr4"], out["data/full/patch_hr"] = pl[0], ph
    # set_channel gray -> 3 channels, 2-D input; np2Tensor with rgb_range 1
    g2 = (np.arange(12 * 10) % 251).astype(np.uint8).reshape(12, 10)
    cl, ch = RD.set_channel([g2], g2, n_channels=3)
    out["data/setchan/in"], out["data/setchan/lr"], out["data/setchan/hr"] = g2, cl[0], ch
    tl, th = RD.np2Tensor([g2[:, :, None]], g2[:, :, None], rgb_range=1)
    out["data/np2tensor/lr"], out["data/np2tensor/hr"] = tl[0].numpy(), th.numpy()
    np.savez_compressed(os.path.join(HERE, "host_golden.npz"), **out)
    print("host_golden.npz", os.path.getsize(os.path.join(HERE, "host_golden.npz")) // 1024, "KiB")


if __name__ == "__main__":
    main()
