"""Loss factory - host-side mirror of reference ``src/loss.py`` (``Loss(opt, ckp)``, 72-152): the ``'w*TYPE+w*TYPE'``
string, the per-epoch loss log (``start_log`` / ``end_log`` / ``display_loss`` / ``save`` -> ``loss_log.pt``) and the four
loss types L1 / MSE / PSNR / SSIM.  Values and gradients come from the engine's reductions (C ABI ``srad_loss_forward`` /
``srad_loss_backward``, csrc/kernels_loss.hip); no torch arithmetic touches the images.

Two ways in:
  * ``loss(sr, hr)`` - the reference's call (src/trainer.py:188): returns a 0-d tensor that carries autograd, so
    ``loss.backward()`` hands ``dLoss/dsr`` to the engine's backward through the model's autograd seam;
  * ``loss.value_and_grad(sr, hr)`` - what the fused training step uses: (0-d loss tensor, dLoss/dsr), no autograd.
The reference reads ``.item()`` of every term every batch (a device sync, loss.py:115); here the log accumulates on the
GPU and is read when ``display_loss`` / ``end_log`` need it.  The PDF plots (loss.py:137-149) are out of scope.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import List, Optional, Tuple

import torch
import torch.nn as nn

from . import _lib as L

KINDS = {"L1": 0, "MSE": 1, "PSNR": 2, "SSIM": 3}


def _workspace(kind: int, shape, device):
    B, Cc, H, W = shape
    nbytes = C.c_size_t()
    L.check(L.lib().srad_loss_workspace_bytes(kind, B, Cc, H, W, C.byref(nbytes)), "loss_workspace_bytes")
    t = torch.empty(nbytes.value + 256, dtype=torch.uint8, device=device)
    off = (-t.data_ptr()) % 256
    return t, C.c_void_p(t.data_ptr() + off), C.c_size_t(nbytes.value)


def _prep(sr: torch.Tensor, hr: torch.Tensor, kind: int):
    if not (sr.is_cuda and hr.is_cuda):
        raise RuntimeError("srad_amd losses run on the GPU only (HIP reductions); there is no CPU fallback")
    if sr.dim() != 4 or hr.dim() != 4 or sr.shape[:2] != hr.shape[:2]:
        raise ValueError(f"expected [B, C, H, W] tensors with equal batch and channels, got {tuple(sr.shape)} and {tuple(hr.shape)}")
    if kind != KINDS["SSIM"] and sr.shape != hr.shape:
        raise ValueError(f"sr {tuple(sr.shape)} and hr {tuple(hr.shape)} differ")
    return sr.detach().float().contiguous(), hr.detach().float().contiguous()


def loss_forward(kind: int, sr, hr, rgb_range: float = 255.0, batch_size: int = 1):
    """-> (out2 [loss, state] float64 on the GPU, workspace handle the backward needs)."""
    sr, hr = _prep(sr, hr, kind)
    B, Cc, sH, sW = sr.shape
    H, W = hr.shape[2:]
    ws = _workspace(kind, hr.shape, sr.device)
    out2 = torch.empty(2, dtype=torch.float64, device=sr.device)
    L.check(L.lib().srad_loss_forward(kind, L.dptr(sr), L.dptr(hr), B, Cc, sH, sW, H, W, float(rgb_range), int(batch_size),
                                      L.dptr(out2), ws[1], ws[2], L.current_stream_ptr()), "loss_forward")
    return out2, (sr, hr, ws)


def loss_backward(kind: int, saved, out2, rgb_range: float, batch_size: int, weight: float = 1.0,
                  gscale: Optional[torch.Tensor] = None, into: Optional[torch.Tensor] = None) -> torch.Tensor:
    sr, hr, ws = saved
    B, Cc, sH, sW = sr.shape
    H, W = hr.shape[2:]
    dsr = into if into is not None else torch.empty_like(sr)
    gs = None if gscale is None else gscale.detach().float().reshape(1).contiguous()
    L.check(L.lib().srad_loss_backward(kind, L.dptr(sr), L.dptr(hr), B, Cc, sH, sW, H, W, float(rgb_range), int(batch_size),
                                       L.dptr(out2), L.dptr(gs), float(weight), L.dptr(dsr), 1 if into is not None else 0,
                                       ws[1], ws[2], L.current_stream_ptr()), "loss_backward")
    return dsr


class _LossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, sr, hr, kind, rgb_range, batch_size):
        out2, saved = loss_forward(kind, sr, hr, rgb_range, batch_size)
        ctx.kind, ctx.rgb_range, ctx.batch_size, ctx.saved, ctx.out2 = kind, rgb_range, batch_size, saved, out2
        return out2[0].float()

    @staticmethod
    def backward(ctx, g):
        dsr = loss_backward(ctx.kind, ctx.saved, ctx.out2, ctx.rgb_range, ctx.batch_size, 1.0, gscale=g)
        return dsr, None, None, None, None


class _Term(nn.Module):
    def __init__(self, kind: str, rgb_range: float = 255.0, batch_size: int = 1):
        super().__init__()
        self.kind, self.rgb_range, self.batch_size = KINDS[kind], float(rgb_range), int(batch_size)

    def forward(self, sr, hr):
        return _LossFn.apply(sr, hr, self.kind, self.rgb_range, self.batch_size)


class L1Loss(_Term):
    """nn.L1Loss(reduction='mean') (src/loss.py:84)."""

    def __init__(self):
        super().__init__("L1")


class MSELoss(_Term):
    """nn.MSELoss() (src/loss.py:82)."""

    def __init__(self):
        super().__init__("MSE")


class PSNRLoss(_Term):
    """src/loss.py:63-70: -10 log10(255^2 / (mse + 1e-8))."""

    def __init__(self):
        super().__init__("PSNR")


class SSIMLoss(_Term):
    """src/loss.py:54-61: calc_ssim(sr, hr, opt.batch_size, 4, opt.rgb_range)."""

    def __init__(self, args):
        super().__init__("SSIM", args.rgb_range, args.batch_size)


class Loss(nn.modules.loss._Loss):
    def __init__(self, args, ckp=None):
        super().__init__()
        self.loss: List[dict] = []
        self.loss_module = nn.ModuleList()
        for term in args.loss.split('+'):
            weight, loss_type = term.split('*')
            if loss_type == 'MSE':
                fn = MSELoss()
            elif loss_type == 'L1':
                fn = L1Loss()
            elif loss_type == 'PSNR':
                fn = PSNRLoss()
            elif loss_type == 'SSIM':
                fn = SSIMLoss(args)
            else:
                assert False, f"Unsupported loss type: {loss_type:s}"
            self.loss.append({'type': loss_type, 'weight': float(weight), 'function': fn})
        if len(self.loss) > 1:
            self.loss.append({'type': 'Total', 'weight': 0, 'function': None})
        for l in self.loss:
            if l['function'] is not None:
                print('{:.3f} * {}'.format(l['weight'], l['type']))
                self.loss_module.append(l['function'])
        self.log = torch.Tensor()
        self._acc: Optional[torch.Tensor] = None           # this epoch's running sums, on the GPU

    # -- evaluation -------------------------------------------------------------------------------------------------
    def note(self, values: List[torch.Tensor]) -> None:
        row = torch.stack([v.detach().double().reshape(()) for v in values])
        if len(self.loss) > 1:
            row = torch.cat([row, row.sum().reshape(1)])
        self._acc = row if self._acc is None else self._acc + row

    def forward(self, sr, hr):
        parts = [l['weight'] * l['function'](sr, hr) for l in self.loss if l['function'] is not None]
        self.note(parts)
        return sum(parts)

    def value_and_grad(self, sr, hr) -> Tuple[torch.Tensor, torch.Tensor]:
        """(sum of the weighted terms, d sum / d sr) without autograd: one forward + one backward launch pair per term,
        the gradients accumulated into one buffer."""
        dsr, parts = None, []
        for l in self.loss:
            fn = l['function']
            if fn is None:
                continue
            out2, saved = loss_forward(fn.kind, sr, hr, fn.rgb_range, fn.batch_size)
            dsr = loss_backward(fn.kind, saved, out2, fn.rgb_range, fn.batch_size, l['weight'], into=dsr)
            parts.append(out2[0] * l['weight'])
        self.note(parts)
        return sum(parts), dsr

    # -- the log (src/loss.py:123-152) ------------------------------------------------------------------------------
    def start_log(self):
        self.log = torch.cat((self.log, torch.zeros(1, len(self.loss))))
        self._acc = None

    def _flush(self):
        """Fold this rank's running sums into the log row.  Under data parallelism every rank has accumulated the loss of ITS
        slice of each global minibatch: the rows are averaged over the ranks first, so the log (rank 0 writes it) is the loss of
        the global minibatch, which is what the reference's single process logs.  Collective: every rank flushes at the same
        points (display_loss / end_log are called by all of them)."""
        if self._acc is not None and self.log.numel():
            acc = self._acc
            try:
                import torch.distributed as dist
                if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
                    acc = acc.clone()
                    dist.all_reduce(acc)
                    acc /= dist.get_world_size()
            except ImportError:
                pass
            self.log[-1] += acc.float().cpu()
            self._acc = None

    def end_log(self, n_batches):
        self._flush()
        self.log[-1].div_(n_batches)

    def display_loss(self, batch):
        self._flush()
        n_samples = batch + 1
        return ''.join('[{}: {:.4f}]'.format(l['type'], c / n_samples) for l, c in zip(self.loss, self.log[-1]))

    def plot_loss(self, apath, epoch):
        """PDF plots are out of scope; the numbers are in loss_log.pt."""

    def save(self, apath):
        self._flush()
        torch.save(self.log, os.path.join(apath, 'loss_log.pt'))
