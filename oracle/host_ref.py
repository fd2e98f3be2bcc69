"""ORACLE (test infrastructure only) - CPU restatement of the reference's loss types and of the sample-preparation
functions of its data loader.  Only ``tests/`` may import this file; the product path never does.

Pinned against the reference itself: ``tests/golden/make_host_golden.py`` imports ``src/loss.py`` and ``src/data.py``
from /root/reference and stores inputs / outputs (loss values, autograd gradients, patches and flips under fixed
``random.seed``s) in ``tests/golden/host_golden.npz``; ``tests/test_oracle_golden.py`` holds this file to them.
Unpinned: ``rgb2y`` (scikit-image's rgb2ycbcr is not importable here) - restated from its published
matrix only.  Each function cites the reference lines it follows (paths relative to /root/reference)."""
from __future__ import annotations

import random
from typing import List, Sequence

import numpy as np
import torch
import torch.nn.functional as F


# ------------------------------------------------------------------ loss types (src/loss.py)
def l1_loss(sr: torch.Tensor, hr: torch.Tensor) -> torch.Tensor:
    """nn.L1Loss(reduction='mean') (src/loss.py:84)."""
    return (sr - hr).abs().mean()


def mse_loss(sr: torch.Tensor, hr: torch.Tensor) -> torch.Tensor:
    """nn.MSELoss() (src/loss.py:82)."""
    return ((sr - hr) ** 2).mean()


def psnr_loss(sr: torch.Tensor, hr: torch.Tensor) -> torch.Tensor:
    """PSNRLoss.forward (src/loss.py:67-70)."""
    return -(10 * torch.log10((255 ** 2) / (mse_loss(sr, hr) + 1e-8)))


def ssim_loss(sr: torch.Tensor, hr: torch.Tensor, batch_size: int, rgb_range: float = 255, scale: int = 4) -> torch.Tensor:
    """calc_ssim (src/loss.py:9-52) as SSIMLoss calls it (scale = 4, src/loss.py:61)."""
    if sr.size(-2) > hr.size(-2) or sr.size(-1) > hr.size(-1):
        sr = sr[:, :, :hr.size(-2), :hr.size(-1)]
    sr = sr.div(rgb_range).clamp(0, 1)
    hr = hr.div(rgb_range).clamp(0, 1)
    shave = scale + 6
    cut = shave if sr.size(-1) > 2 * shave else 1
    sr, hr = sr[..., cut:-cut, cut:-cut], hr[..., cut:-cut, cut:-cut]
    if sr.size(1) > 1:
        w = torch.tensor([65.738, 129.057, 25.064], dtype=sr.dtype).view(1, 3, 1, 1) / 256
        sr, hr = (sr * w).sum(1, keepdim=True), (hr * w).sum(1, keepdim=True)
    c1, c2 = (0.01 * 255) ** 2, (0.03 * 255) ** 2
    k = torch.ones(1, 1, 11, 11, dtype=sr.dtype) / 121
    box = lambda t: F.conv2d(t, k, padding=5)
    m1, m2 = box(sr), box(hr)
    s1, s2, s12 = box(sr * sr) - m1 * m1, box(hr * hr) - m2 * m2, box(sr * hr) - m1 * m2
    smap = ((2 * m1 * m2 + c1) * (2 * s12 + c2)) / ((m1 * m1 + m2 * m2 + c1) * (s1 + s2 + c2))
    return (1 - smap).sum() / batch_size


def total_loss(spec: str, sr: torch.Tensor, hr: torch.Tensor, batch_size: int = 1, rgb_range: float = 255) -> torch.Tensor:
    """Loss.forward (src/loss.py:108-121): sum of weight * term over the 'w*TYPE+w*TYPE' string."""
    total = 0
    for term in spec.split('+'):
        w, kind = term.split('*')
        fn = {"L1": l1_loss, "MSE": mse_loss, "PSNR": psnr_loss,
              "SSIM": lambda a, b: ssim_loss(a, b, batch_size, rgb_range)}[kind]
        total = total + float(w) * fn(sr, hr)
    return total


# ------------------------------------------------------------------ sample preparation (src/data.py)
def get_patch(lrs: Sequence[np.ndarray], hr: np.ndarray, patch_size: int, scale: Sequence[int], rng=random):
    """src/data.py:21-36; ``scale`` coarsest first, like SRData.scale."""
    th, tw = hr.shape[:2]
    tx = rng.randrange(0, tw - patch_size + 1)
    ty = rng.randrange(0, th - patch_size + 1)
    tx, ty = tx - tx % scale[0], ty - ty % scale[0]
    out = [lrs[i][ty // s:ty // s + patch_size // s, tx // s:tx // s + patch_size // s, :] for i, s in enumerate(scale)]
    return out, hr[ty:ty + patch_size, tx:tx + patch_size, :]


def augment(lrs: Sequence[np.ndarray], hr: np.ndarray, rng=random):
    """src/data.py:38-50 with hflip = rot = True."""
    h, v, t = rng.random() < 0.5, rng.random() < 0.5, rng.random() < 0.5

    def f(a):
        if h:
            a = a[:, ::-1, :]
        if v:
            a = a[::-1, :, :]
        if t:
            a = a.transpose(1, 0, 2)
        return a
    return [f(a) for a in lrs], f(hr)


def rgb2y(img: np.ndarray) -> np.ndarray:
    """skimage.color.rgb2ycbcr(img)[:, :, 0] for uint8 RGB (published matrix; PARITY UNPINNED: skimage is absent)."""
    return 16.0 + (img.astype(np.float64) / 255.0) @ np.array([65.481, 128.553, 24.966])


def virtual_index(idx: int, n_images: int, test_every: int, batch_size: int):
    """SRData._get_index for training (src/data.py:101-105,148-155): (image index or None = uniformly random, length)."""
    length = test_every * batch_size
    border = n_images * (length // n_images)
    return (idx % n_images if idx < border else None), length
