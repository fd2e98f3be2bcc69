"""Trainer for DRCT and DRN-L (+ dual regression models) on the HIP engine - the loop of reference
src/trainer.py:152-222 (train) and 242-304 (test) with the reference's optimizer / scheduler settings (49-83).
Differences, all deliberate:
  * the optimizer is the fused Adam kernel on the flat parameter buffer (same arithmetic as torch.optim.Adam);
  * no fp16 autocast / GradScaler: the engine's bf16 mode keeps fp32 master weights, fp32 accumulation and fp32
    gradients, which needs no loss scaling;
  * under torch.distributed every rank takes its slice of each minibatch and gradients are all-reduced per RDG
    bucket while the backward is still running (BASELINE config C4); DRN gradients are all-reduced after the backward."""
from __future__ import annotations

import os
import time
from pathlib import Path
from typing import Iterable, List, Sequence, Tuple

import numpy as np
import torch

from . import metrics as M
from .train import FusedAdam, GradReducer, GraphedTrainStep, cosine_lr, drn_train_step, train_step


def make_optimizer(opt, my_model) -> FusedAdam:
    """src/trainer.py:49-59."""
    return FusedAdam(my_model.get_model() if hasattr(my_model, "get_model") else my_model, lr=opt.lr,
                     betas=(opt.beta1, opt.beta2), eps=opt.epsilon, weight_decay=opt.weight_decay)


class CosineSchedule:
    """lrs.CosineAnnealingLR(optimizer, float(opt.epochs), eta_min=opt.eta_min) stepped once per epoch
    (src/trainer.py:76-83, 224-228)."""

    def __init__(self, optimizer: FusedAdam, epochs: float, eta_min: float):
        self.optimizer, self.t_max, self.eta_min = optimizer, float(epochs), float(eta_min)
        self.base_lr = float(optimizer.param_groups[0]["lr"])
        self.last_epoch = 0

    def get_last_lr(self) -> List[float]:
        return [float(self.optimizer.param_groups[0]["lr"])]

    def step(self) -> None:
        self.last_epoch += 1
        self.optimizer.param_groups[0]["lr"] = cosine_lr(self.base_lr, self.last_epoch, self.t_max, self.eta_min)


def make_scheduler(opt, my_optimizer) -> CosineSchedule:
    return CosineSchedule(my_optimizer, float(opt.epochs), opt.eta_min)


# ------------------------------------------------------------------ data: {root}/{class}/train/good/{HR, LR_s}/*.png
def _load_png(path: str, n_colors: int) -> np.ndarray:
    from PIL import Image
    im = Image.open(path)
    im = im.convert("L" if n_colors == 1 else "RGB")
    a = np.asarray(im, dtype=np.uint8)
    return a[:, :, None] if a.ndim == 2 else a


class FolderPairs:
    """Training pairs in the layout scripts/prepare_mvtec_data.py writes and src/data.py:109-147 reads: HR images in
    ``{dir}/HR`` and the LR twins in ``{dir}/LR_{scale}`` (or ``LR_bicubic/X{scale}``, ``LR``).  Images are kept as u8
    arrays in host memory (MVTec classes are a few hundred 128 px tiles)."""

    def __init__(self, data_dir: str, scale, n_colors: int):
        """``scale``: one factor (DRCT) or the list ``opt.scale`` = [2, 4, ..] of DRN, whose loader yields the LR image
        of every scale, coarsest first (src/data.py:109-147, src/trainer.py:161)."""
        base = Path(data_dir)
        hr_dir = base / "HR"
        scales = sorted(scale, reverse=True) if isinstance(scale, (list, tuple)) else [scale]
        self.multi = isinstance(scale, (list, tuple))
        lr_dirs = []
        for sc in scales:
            d = next((d for d in (base / f"LR_{sc}", base / "LR_bicubic" / f"X{sc}", base / "LR") if d.is_dir()), None)
            if d is None:
                raise FileNotFoundError(f"expected an LR folder for scale {sc} next to {hr_dir}")
            lr_dirs.append(d)
        if not hr_dir.is_dir():
            raise FileNotFoundError(f"expected {hr_dir}")
        self.items: List[Tuple[str, object, np.ndarray]] = []
        for hp in sorted(hr_dir.glob("*.png")):
            lrs = []
            for d in lr_dirs:
                lp = d / hp.name
                if not lp.is_file():
                    raise FileNotFoundError(str(lp))
                lrs.append(_load_png(str(lp), n_colors))
            self.items.append((hp.stem, lrs if self.multi else lrs[0], _load_png(str(hp), n_colors)))
        if not self.items:
            raise FileNotFoundError(f"no PNG files under {hr_dir}")

    def __len__(self) -> int:
        return len(self.items)


def batches(ds: FolderPairs, batch_size: int, epoch: int, rank: int = 0, world: int = 1, shuffle: bool = True,
            augment: bool = False, seed: int = 1) -> Iterable[Tuple[torch.Tensor, torch.Tensor, Sequence[str]]]:
    """Minibatches of (lr, hr, names) as fp32 NCHW tensors in [0, 255] on the host.  Every rank draws the same
    permutation and takes the slice ``rank::world`` of each global batch (so the union over ranks is the
    reference's batch).  ``augment``: random h/v flips + transpose (src/data.py augment), per image."""
    g = np.random.default_rng(seed + epoch)
    order = g.permutation(len(ds)) if shuffle else np.arange(len(ds))
    for i in range(0, len(order) - batch_size + 1, batch_size):
        idx = order[i:i + batch_size][rank::world]
        if len(idx) == 0:
            continue
        def aug(a, f):
            if f[0]:
                a = a[:, ::-1]
            if f[1]:
                a = a[::-1]
            if f[2]:
                a = a.transpose(1, 0, 2)
            return np.ascontiguousarray(a)

        def to_t(arrs):
            return torch.from_numpy(np.stack(arrs)).permute(0, 3, 1, 2).float().contiguous()
        multi = getattr(ds, "multi", False)
        lrs, hrs, names = [], [], []
        for j in idx:
            name, lr, hr = ds.items[int(j)]
            f = g.integers(0, 2, size=3) if augment else (0, 0, 0)
            lrs.append([aug(a, f) for a in lr] if multi else aug(lr, f))
            hrs.append(aug(hr, f))
            names.append(name)
        lr_t = [to_t([l[k] for l in lrs]) for k in range(len(lrs[0]))] if multi else to_t(lrs)
        yield lr_t, to_t(hrs), names


class Trainer:
    """src/trainer.py:117-305 for ``--model-type drct``: ``train()`` runs one epoch, ``test()`` the validation
    PSNR / SSIM of Trainer.test, ``terminate()`` the epoch budget."""

    def __init__(self, opt, train_set: FolderPairs, my_model, ckp=None, val_set: FolderPairs = None, dual_model: bool = False):
        import torch.distributed as dist
        self.opt = opt
        self.scale = opt.scale
        self.ckp = ckp
        self.model = my_model
        self.dual_model = dual_model
        self.net = my_model.get_model() if hasattr(my_model, "get_model") else my_model
        if not self.net._can_train():
            raise NotImplementedError(f"{type(self.net).__name__}: this configuration is inference-only on the HIP engine "
                                      "(DRN x8: n_feats = 10)")
        self.net.train()
        self.net.enable_training()
        self.train_set, self.val_set = train_set, val_set
        self.optimizer = make_optimizer(opt, self.net)
        self.scheduler = make_scheduler(opt, self.optimizer)
        if dual_model:                                       # src/trainer.py:62-73, 86-96, 126-129
            self.dual_models = my_model.dual_models
            self.dual_optimizers = [torch.optim.Adam(dm.parameters(), lr=opt.lr, betas=(opt.beta1, opt.beta2), eps=opt.epsilon,
                                                     weight_decay=opt.weight_decay) for dm in self.dual_models]
            self.dual_scheduler = [torch.optim.lr_scheduler.CosineAnnealingLR(o, float(opt.epochs), eta_min=opt.eta_min)
                                   for o in self.dual_optimizers]
        self.rank = dist.get_rank() if dist.is_initialized() else 0
        self.world = dist.get_world_size() if dist.is_initialized() else 1
        self.reducer = GradReducer().attach(self.net) if self.world > 1 else None
        # one GPU, DRCT: the whole step is one replayed hipGraph per batch shape (train.GraphedTrainStep); opt.train_graph = False
        # keeps the eager launches.  Data parallel runs stay eager: the bucket hooks launch RCCL all-reduces mid-backward.
        self.graph_step = (GraphedTrainStep(self.net, self.optimizer)
                           if self.world == 1 and not dual_model and getattr(opt, "train_graph", True) else None)
        self.error_last = 1e8
        self.loss_log: List[float] = []
        self.device = next(self.net.parameters()).device

    def _log(self, msg: str) -> None:
        if self.rank == 0:
            (self.ckp.write_log(msg) if self.ckp is not None else print(msg))

    def train(self) -> float:
        epoch = self.scheduler.last_epoch + 1
        self._log('[Epoch {}]\tLearning rate: {:.2e}'.format(epoch, self.scheduler.get_last_lr()[0]))
        self.net.train()
        losses = []
        t0 = time.perf_counter()
        bs = self.opt.batch_size
        for batch, (lr, hr, _) in enumerate(batches(self.train_set, bs, epoch, self.rank, self.world,
                                                    augment=not getattr(self.opt, "no_augment", True))):
            hr = hr.to(self.device, non_blocking=True)
            if self.dual_model:
                lr = [a.to(self.device, non_blocking=True) for a in lr]
                losses.append(drn_train_step(self.net, self.dual_models, lr, hr, self.optimizer, self.dual_optimizers,
                                             getattr(self.opt, "dual_weight", 0.1), self.reducer))
            else:
                lr = lr.to(self.device, non_blocking=True)
                losses.append(self.graph_step(lr, hr) if self.graph_step is not None else
                              train_step(self.net, lr, hr, self.optimizer, self.reducer))
            if (batch + 1) % self.opt.print_every == 0:
                cur = float(torch.stack(losses[-self.opt.print_every:]).mean())      # the only host sync, every print_every
                self._log('[{}/{}]\t[L1: {:.4f}]\t{:.1f}s'.format((batch + 1) * bs, len(self.train_set), cur,
                                                                   time.perf_counter() - t0))
        mean = float(torch.stack(losses).mean()) if losses else float("nan")
        self.loss_log.append(mean)
        self.error_last = mean
        self.scheduler.step()
        if self.dual_model:
            for sch in self.dual_scheduler:
                sch.step()
        return mean

    @torch.no_grad()
    def test(self) -> Tuple[float, float]:
        """Validation PSNR / SSIM (src/trainer.py:242-304: eval mode, quantize, 4-px shave, the 255^2 SSIM constants)."""
        if self.val_set is None:
            return float("nan"), float("nan")
        self.net.eval()
        ps, ss = [], []
        for lr, hr, _ in batches(self.val_set, 1, 0, shuffle=False):
            sr = self.model((lr[0] if isinstance(lr, list) else lr).to(self.device))
            if isinstance(sr, (list, tuple)):
                sr = sr[-1]                                   # src/trainer.py:269
            sr = M.quantize(sr, self.opt.rgb_range)
            p, s = M.val_metrics(sr, hr.to(self.device), self.opt.rgb_range)
            ps.append(float(p.mean()))
            ss.append(float(s.mean()))
        self.net.train()
        return float(np.mean(ps)), float(np.mean(ss))

    def terminate(self) -> bool:
        return self.scheduler.last_epoch >= self.opt.epochs


def train_drct(opt, ckp=None) -> dict:
    """src/main.py train_drct (327-388): build the model, train ``opt.epochs`` epochs, keep model_latest / model_best
    under ``opt.save/model``.  Returns the loss per epoch."""
    from .model import Model
    model = Model(opt, ckp)
    train_set = FolderPairs(opt.data_dir, max(opt.scale), opt.n_colors)
    t = Trainer(opt, train_set, model, ckp)
    best = float("inf")
    while not t.terminate():
        loss = t.train()
        t._log(f"epoch {t.scheduler.last_epoch}: mean L1 {loss:.4f}")
        if t.rank == 0:
            model.save(opt.save, is_best=loss < best)
        best = min(best, loss)
    return {"loss": t.loss_log}


def train_drn(opt, ckp=None) -> dict:
    """src/main.py train_drn (292-325): DRN-L with one dual regression model per scale."""
    from .model import Model
    model = Model(opt, ckp, dual_model=True)
    train_set = FolderPairs(opt.data_dir, list(opt.scale), opt.n_colors)
    t = Trainer(opt, train_set, model, ckp, dual_model=True)
    best = float("inf")
    while not t.terminate():
        loss = t.train()
        t._log(f"epoch {t.scheduler.last_epoch}: mean loss {loss:.4f}")
        if t.rank == 0:
            model.save(opt.save, is_best=loss < best)
        best = min(best, loss)
    return {"loss": t.loss_log}
