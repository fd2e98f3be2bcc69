"""Trainer for the DRCT model on the HIP engine - the loop of reference src/trainer.py:152-222 (train) and
242-304 (test) with the reference's optimizer / scheduler settings (49-83).  Differences, all deliberate:
  * the optimizer is the fused Adam kernel on the flat parameter buffer (same arithmetic as torch.optim.Adam);
  * no fp16 autocast / GradScaler: the engine's bf16 mode keeps fp32 master weights, fp32 accumulation and fp32
    gradients, which needs no loss scaling;
  * under torch.distributed every rank takes its slice of each minibatch and gradients are all-reduced per RDG
    bucket while the backward is still running (BASELINE config C4).
DRN training (dual regression, src/trainer.py:168-185) needs the DRN backward, which is not built yet."""
from __future__ import annotations

import os
import time
from pathlib import Path
from typing import Iterable, List, Sequence, Tuple

import numpy as np
import torch

from . import metrics as M
from .train import FusedAdam, GradReducer, cosine_lr, train_step


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

    def __init__(self, data_dir: str, scale: int, n_colors: int):
        base = Path(data_dir)
        hr_dir = base / "HR"
        lr_dir = next((d for d in (base / f"LR_{scale}", base / "LR_bicubic" / f"X{scale}", base / "LR") if d.is_dir()), None)
        if not hr_dir.is_dir() or lr_dir is None:
            raise FileNotFoundError(f"expected {hr_dir} and an LR folder next to it")
        self.items: List[Tuple[str, np.ndarray, np.ndarray]] = []
        for hp in sorted(hr_dir.glob("*.png")):
            lp = lr_dir / hp.name
            if not lp.is_file():
                raise FileNotFoundError(str(lp))
            self.items.append((hp.stem, _load_png(str(lp), n_colors), _load_png(str(hp), n_colors)))
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
        lrs, hrs, names = [], [], []
        for j in idx:
            name, lr, hr = ds.items[int(j)]
            if augment:
                f = g.integers(0, 2, size=3)
                if f[0]:
                    lr, hr = lr[:, ::-1], hr[:, ::-1]
                if f[1]:
                    lr, hr = lr[::-1], hr[::-1]
                if f[2]:
                    lr, hr = lr.transpose(1, 0, 2), hr.transpose(1, 0, 2)
            lrs.append(np.ascontiguousarray(lr))
            hrs.append(np.ascontiguousarray(hr))
            names.append(name)
        yield (torch.from_numpy(np.stack(lrs)).permute(0, 3, 1, 2).float().contiguous(),
               torch.from_numpy(np.stack(hrs)).permute(0, 3, 1, 2).float().contiguous(), names)


class Trainer:
    """src/trainer.py:117-305 for ``--model-type drct``: ``train()`` runs one epoch, ``test()`` the validation
    PSNR / SSIM of Trainer.test, ``terminate()`` the epoch budget."""

    def __init__(self, opt, train_set: FolderPairs, my_model, ckp=None, val_set: FolderPairs = None):
        import torch.distributed as dist
        self.opt = opt
        self.scale = opt.scale
        self.ckp = ckp
        self.model = my_model
        self.net = my_model.get_model() if hasattr(my_model, "get_model") else my_model
        if not self.net._can_train():
            raise NotImplementedError(f"{type(self.net).__name__}: training on the HIP engine is built for DRCT only")
        self.net.train()
        self.net.enable_training()
        self.train_set, self.val_set = train_set, val_set
        self.optimizer = make_optimizer(opt, self.net)
        self.scheduler = make_scheduler(opt, self.optimizer)
        self.rank = dist.get_rank() if dist.is_initialized() else 0
        self.world = dist.get_world_size() if dist.is_initialized() else 1
        self.reducer = GradReducer().attach(self.net) if self.world > 1 else None
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
            lr, hr = lr.to(self.device, non_blocking=True), hr.to(self.device, non_blocking=True)
            losses.append(train_step(self.net, lr, hr, self.optimizer, self.reducer))
            if (batch + 1) % self.opt.print_every == 0:
                cur = float(torch.stack(losses[-self.opt.print_every:]).mean())      # the only host sync, every print_every
                self._log('[{}/{}]\t[L1: {:.4f}]\t{:.1f}s'.format((batch + 1) * bs, len(self.train_set), cur,
                                                                   time.perf_counter() - t0))
        mean = float(torch.stack(losses).mean()) if losses else float("nan")
        self.loss_log.append(mean)
        self.error_last = mean
        self.scheduler.step()
        return mean

    @torch.no_grad()
    def test(self) -> Tuple[float, float]:
        """Validation PSNR / SSIM (src/trainer.py:242-304: eval mode, quantize, 4-px shave, the 255^2 SSIM constants)."""
        if self.val_set is None:
            return float("nan"), float("nan")
        self.net.eval()
        ps, ss = [], []
        for lr, hr, _ in batches(self.val_set, 1, 0, shuffle=False):
            sr = self.model(lr.to(self.device))
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
