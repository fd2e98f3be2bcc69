"""Reconstruction-error scorer on the GPU - host side of the C ABI's scorer entry points
(include/srad.h), mirroring reference src/metrics.py, the u8 conversions of src/evaluate.py:214-215
and src/trainer.py:45-47, and ``sklearn.metrics.roc_auc_score`` as src/evaluate.py uses it.
No CPU fallback: tensors must live on the GPU."""
from __future__ import annotations

import ctypes as C
from typing import List, Optional, Sequence, Tuple

import numpy as np
import torch

from . import _lib as L


def _ws_buffer(nbytes: int, device) -> Tuple[torch.Tensor, C.c_void_p, C.c_size_t]:
    t = torch.empty(nbytes + 256, dtype=torch.uint8, device=device)
    off = (-t.data_ptr()) % 256
    return t, C.c_void_p(t.data_ptr() + off), C.c_size_t(nbytes)


def _need_cuda(*ts):
    for t in ts:
        if not t.is_cuda:
            raise RuntimeError("srad_amd.metrics runs on the GPU only (HIP scorer); there is no CPU fallback")


def to_u8_hwc(x: torch.Tensor, rgb_range: float = 255.0) -> torch.Tensor:
    """``x.mul(255/range).clamp(0,255).byte().permute(0,2,3,1)`` - TRUNCATION, as the evaluator
    converts images (src/evaluate.py:214-215).  x: [B,C,H,W] fp32 -> [B,H,W,C] uint8."""
    _need_cuda(x)
    x = x.detach().float().contiguous()
    B, Cc, H, W = x.shape
    out = torch.empty(B, H, W, Cc, dtype=torch.uint8, device=x.device)
    L.check(L.lib().srad_to_u8_hwc(L.dptr(x), B, Cc, H, W, float(rgb_range), L.dptr(out), L.current_stream_ptr()), "to_u8_hwc")
    return out


def quantize(x: torch.Tensor, rgb_range: float = 255.0) -> torch.Tensor:
    """``quantize`` of the trainer (src/trainer.py:45-47): mul, clamp, ROUND (half to even), div."""
    _need_cuda(x)
    x = x.detach().float().contiguous()
    y = torch.empty_like(x)
    L.check(L.lib().srad_quantize(L.dptr(x), L.dptr(y), C.c_int64(x.numel()), float(rgb_range), L.current_stream_ptr()), "quantize")
    return y


def score_pairs(sr_u8: torch.Tensor, hr_u8: torch.Tensor, window_sizes: Sequence[int]
                ) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    """Per-pair scores of [n,H,W,C] uint8 image stacks (SR, HR):
    ssim[n, len(window_sizes)] = ssim_numpy(hr/255, sr/255, ws) (src/metrics.py:26-67), mse[n], psnr[n]
    (float64 tensors on the GPU)."""
    _need_cuda(sr_u8, hr_u8)
    assert sr_u8.dtype == torch.uint8 and hr_u8.dtype == torch.uint8 and sr_u8.shape == hr_u8.shape and sr_u8.dim() == 4
    sr_u8, hr_u8 = sr_u8.contiguous(), hr_u8.contiguous()
    n, H, W, Cc = sr_u8.shape
    ws = (C.c_int32 * max(1, len(window_sizes)))(*[int(w) for w in window_sizes])
    dev = sr_u8.device
    ssim = torch.empty(n, len(window_sizes), dtype=torch.float64, device=dev)
    mse = torch.empty(n, dtype=torch.float64, device=dev)
    psnr = torch.empty(n, dtype=torch.float64, device=dev)
    nbytes = C.c_size_t()
    L.check(L.lib().srad_score_workspace_bytes(n, H, W, C.byref(nbytes)), "score_workspace_bytes")
    keep, wp, wb = _ws_buffer(nbytes.value, dev)
    L.check(L.lib().srad_score_pairs(L.dptr(sr_u8), L.dptr(hr_u8), n, H, W, Cc, ws, len(window_sizes), L.dptr(ssim),
                                     L.dptr(mse), L.dptr(psnr), wp, wb, L.current_stream_ptr()), "score_pairs")
    return ssim, mse, psnr


def val_metrics(sr: torch.Tensor, hr: torch.Tensor, rgb_range: float) -> Tuple[torch.Tensor, torch.Tensor]:
    """psnr_torch / ssim_torch of Trainer.test (src/metrics.py:70-108) per image, with the
    reference's quirks kept verbatim: 4-px shave, zero padding, C1/C2 scaled by 255^2."""
    _need_cuda(sr, hr)
    sr, hr = sr.detach().float().contiguous(), hr.detach().float().contiguous()
    if sr.shape[-2] > hr.shape[-2] or sr.shape[-1] > hr.shape[-1]:
        sr = sr[..., :hr.shape[-2], :hr.shape[-1]].contiguous()
    assert sr.shape == hr.shape
    B, Cc, H, W = sr.shape
    psnr = torch.empty(B, dtype=torch.float64, device=sr.device)
    ssim = torch.empty(B, dtype=torch.float64, device=sr.device)
    L.check(L.lib().srad_val_metrics(L.dptr(sr), L.dptr(hr), B, Cc, H, W, float(rgb_range), L.dptr(psnr), L.dptr(ssim),
                                     None, C.c_size_t(0), L.current_stream_ptr()), "val_metrics")
    return psnr, ssim


def roc_auc(y_true: Sequence[int], scores: Sequence[float]) -> float:
    """Binary ROC-AUC, equal to ``sklearn.metrics.roc_auc_score`` (ties count one half)."""
    y = np.ascontiguousarray(np.asarray(y_true), dtype=np.int32)
    s = np.ascontiguousarray(np.asarray(scores), dtype=np.float64)
    if y.shape != s.shape or y.ndim != 1:
        raise ValueError("y_true and scores must be 1-D and of equal length")
    out = C.c_double()
    rc = L.lib().srad_roc_auc(y.ctypes.data_as(C.POINTER(C.c_int32)), s.ctypes.data_as(C.POINTER(C.c_double)), len(y), C.byref(out))
    if rc:
        raise ValueError(L.lib().srad_last_error().decode())
    return out.value


def l1_loss(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    """nn.L1Loss(reduction='mean') (src/loss.py:84) -> 0-d float64 tensor on the GPU."""
    _need_cuda(a, b)
    a, b = a.detach().float().contiguous(), b.detach().float().contiguous()
    assert a.shape == b.shape
    out = torch.empty((), dtype=torch.float64, device=a.device)
    nbytes = C.c_size_t()
    L.check(L.lib().srad_l1_workspace_bytes(C.byref(nbytes)), "l1_workspace_bytes")
    keep, wp, _ = _ws_buffer(nbytes.value, a.device)
    L.check(L.lib().srad_l1_loss(L.dptr(a), L.dptr(b), C.c_int64(a.numel()), L.dptr(out), wp, L.current_stream_ptr()), "l1_loss")
    return out


# ------------------------------------------------------------------ reference-named conveniences
def sweep_window_sizes(min_dim: int) -> List[int]:
    """Window sizes the evaluator sweeps (src/evaluate.py:231-233)."""
    max_w = max(3, min_dim - 3)
    return [w for w in range(3, max_w + 1, 10) if w % 2 == 1] or [3]


def _as_u8_stack(img) -> torch.Tensor:
    a = np.asarray(img)
    if a.dtype != np.uint8:
        q = np.rint(a.astype(np.float64) * 255.0)
        if not np.allclose(q / 255.0, a, atol=1e-6):
            raise ValueError("ssim_numpy/psnr_numpy here take uint8 images or floats that are exact multiples of 1/255 "
                             "(what src/evaluate.py passes); use score_pairs for anything else")
        a = q.astype(np.uint8)
    if a.ndim == 2:
        a = a[:, :, None]
    return torch.from_numpy(np.ascontiguousarray(a))[None].cuda()


def ssim_numpy(img_ref, img, win_size: int = 11) -> float:
    """Drop-in for ``src.metrics.ssim_numpy(hr/255, sr/255, ws)`` as the evaluator calls it."""
    s, _, _ = score_pairs(_as_u8_stack(img), _as_u8_stack(img_ref), [win_size])
    return float(s[0, 0].item())


def psnr_numpy(img_ref, img) -> float:
    """Drop-in for ``src.metrics.psnr_numpy(hr/255, sr/255)``."""
    _, _, p = score_pairs(_as_u8_stack(img), _as_u8_stack(img_ref), [])
    return float(p[0].item())


def psnr_torch(sr: torch.Tensor, hr: torch.Tensor, rgb_range: float) -> float:
    """src/metrics.py:70-79 (batch of one image, as Trainer.test calls it)."""
    p, _ = val_metrics(sr, hr, rgb_range)
    return float(p.mean().item()) if p.numel() > 1 else float(p[0].item())


def ssim_torch(sr: torch.Tensor, hr: torch.Tensor, rgb_range: float, win_size: int = 11) -> float:
    """src/metrics.py:82-108."""
    if win_size != 11:
        raise ValueError("the validation SSIM kernel is built for the 11x11 window the trainer uses")
    _, s = val_metrics(sr, hr, rgb_range)
    return float(s.mean().item()) if s.numel() > 1 else float(s[0].item())


def evaluate_pairs(y_true: Sequence[int], sr_u8: torch.Tensor, hr_u8: torch.Tensor) -> dict:
    """Window sweep + final scores + the three AUCs of ``evaluate_on_test`` (src/evaluate.py:226-267)
    for image stacks already on the GPU.  One scorer launch sequence covers every window size."""
    n, H, W, _ = hr_u8.shape
    sizes = sweep_window_sizes(min(H, W))
    ssim, mse, psnr = score_pairs(sr_u8, hr_u8, sizes)
    ssim_h, mse_h, psnr_h = ssim.cpu().numpy(), mse.cpu().numpy(), psnr.cpu().numpy()
    best_ws, best_auc, sweep, best_j = sizes[0], -1.0, [], 0
    for j, ws in enumerate(sizes):
        a = roc_auc(y_true, 1.0 - ssim_h[:, j])
        sweep.append(a)
        if a > best_auc:                      # strict '>': the first maximum wins (evaluate.py:246)
            best_auc, best_ws, best_j = a, ws, j
    s_ssim = (1.0 - ssim_h[:, best_j]).tolist()
    return dict(window_sizes=sizes, sweep_auc=sweep, best_ws=best_ws,
                scores_ssim=s_ssim, scores_mse=mse_h.tolist(), scores_psnr=psnr_h.tolist(),
                auc_ssim=roc_auc(y_true, s_ssim), auc_mse=roc_auc(y_true, mse_h),
                auc_psnr=roc_auc(y_true, -psnr_h))
