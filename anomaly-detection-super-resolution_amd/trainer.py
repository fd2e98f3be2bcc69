"""Trainer for DRCT and DRN-L (+ dual regression models) on the HIP engine: host-side mirror of reference
``src/trainer.py`` - ``Trainer(opt, loader, my_model, my_loss, ckp, dual_model)`` with ``train()`` (141-227), ``test()``
(242-304), ``step()`` / ``prepare()`` / ``terminate()`` (312-340), ``make_optimizer`` / ``make_scheduler`` and their dual
twins (49-96), ``quantize`` (45-47).  Same log lines, same epoch structure (one epoch = the loader's
``test_every * batch_size`` virtual samples, cosine schedule stepped once per epoch).

Differences, all deliberate:
  * the optimizers are the engine's fused Adam (``train.FusedAdam`` on the flat parameter buffer, ``train.TensorAdam``
    for the two-tensor dual models) - torch.optim.Adam's arithmetic, no torch optimizer on the path;
  * no fp16 autocast / GradScaler (src/trainer.py:128-129,164): the engine's bf16 mode keeps fp32 master weights, fp32
    accumulation and fp32 gradients, which needs no loss scaling;
  * DRCT with the '1*L1' loss of every CLI path runs the whole step as one replayed hipGraph per batch shape on one
    GPU (``train.GraphedTrainStep``); any other loss string runs the same step eagerly through ``Loss.value_and_grad``;
  * under ``torch.distributed`` the loader hands every rank ``rank::world`` of each global minibatch and the gradients
    are all-reduced per RDG bucket while the backward is still running (BASELINE config C4); DRN gradients are
    all-reduced after the backward;
  * the per-batch ``loss.item()`` sync of the reference (src/loss.py:115) is gone: the loss log accumulates on the GPU
    and is read every ``print_every`` batches."""
from __future__ import annotations

import time
from decimal import Decimal
from typing import List

import torch

from . import metrics as M
from .train import (FusedAdam, GradReducer, GraphedDrnTrainStep, GraphedTrainStep, TensorAdam, cosine_lr, drn_train_step, train_step)


class timer():
    """Stopwatch with an accumulator - the interface of the reference's timer (src/trainer.py:21-42): ``tic`` restarts the
    lap, ``toc`` reads it, ``hold`` adds the lap to ``acc``, ``release`` returns ``acc`` and clears it."""
    __slots__ = ("acc", "t0")

    def __init__(self):
        self.acc, self.t0 = 0.0, time.perf_counter()

    def tic(self) -> None:
        self.t0 = time.perf_counter()

    def toc(self) -> float:
        return time.perf_counter() - self.t0

    def hold(self) -> None:
        self.acc += self.toc()

    def release(self) -> float:
        held, self.acc = self.acc, 0.0
        return held

    def reset(self) -> None:
        self.acc = 0.0


def quantize(img, rgb_range):
    """src/trainer.py:45-47: mul, clamp, ROUND, div - on the engine (``srad_quantize``)."""
    return M.quantize(img, rgb_range)


def make_optimizer(opt, my_model) -> FusedAdam:
    """src/trainer.py:49-59."""
    return FusedAdam(my_model.get_model() if hasattr(my_model, "get_model") else my_model, lr=opt.lr,
                     betas=(opt.beta1, opt.beta2), eps=opt.epsilon, weight_decay=opt.weight_decay)


def make_dual_optimizer(opt, dual_models) -> List[TensorAdam]:
    """src/trainer.py:62-73: one Adam per dual model."""
    return [TensorAdam(dm.parameters(), lr=opt.lr, betas=(opt.beta1, opt.beta2), eps=opt.epsilon, weight_decay=opt.weight_decay)
            for dm in dual_models]


class CosineSchedule:
    """lrs.CosineAnnealingLR(optimizer, float(opt.epochs), eta_min=opt.eta_min) stepped once per epoch
    (src/trainer.py:76-83, 224-228)."""

    def __init__(self, optimizer, epochs: float, eta_min: float):
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


def make_dual_scheduler(opt, dual_optimizers) -> List[CosineSchedule]:
    """src/trainer.py:86-96."""
    return [CosineSchedule(o, float(opt.epochs), opt.eta_min) for o in dual_optimizers]


class _PrintLog:
    """Stand-in when no Checkpoint is given (tests, tools): log lines go to stdout, nothing is written."""
    log = torch.zeros(0, 2)

    def write_log(self, msg, refresh=False):
        print(msg)

    def add_log(self, log):
        self.log = torch.cat([self.log, log])

    def save_results_nopostfix(self, filename, sr, scale):
        pass


def _plain_l1(loss) -> bool:
    terms = [l for l in getattr(loss, "loss", []) if l["function"] is not None]
    return loss is None or (len(terms) == 1 and terms[0]["type"] == "L1" and terms[0]["weight"] == 1.0)


class Trainer():
    def __init__(self, opt, loader, my_model, my_loss, ckp, dual_model=False):
        import torch.distributed as dist
        self.opt = opt
        self.scale = opt.scale
        self.ckp = ckp if ckp is not None else _PrintLog()
        self.dual_model = dual_model
        self.loader_train = loader.loader_train
        self.loader_test = loader.loader_test
        self.model = my_model
        self.loss = my_loss
        self.net = my_model.get_model() if hasattr(my_model, "get_model") else my_model
        self.rank = dist.get_rank() if dist.is_initialized() else 0
        self.world = dist.get_world_size() if dist.is_initialized() else 1
        if not getattr(opt, "test_only", False):
            if not self.net._can_train():
                raise NotImplementedError(f"{type(self.net).__name__}: this configuration is inference-only on the HIP engine "
                                          "(DRCT: windows larger than 16 x 16, i.e. resolution / scale > 64)")
            if self.world > 1 and (opt.batch_size % self.world or opt.batch_size < self.world):
                # a rank with an empty slice would skip the step and leave its peers waiting in the bucket all-reduce
                raise ValueError(f"--batch-size {opt.batch_size} is the GLOBAL minibatch: it must be a positive multiple of the "
                                 f"{self.world} ranks it is sharded over")
            if self.loader_train is not None and (getattr(self.loader_train, "world", 1), getattr(self.loader_train, "rank", 0)) != (self.world, self.rank):
                raise ValueError("the training loader was built for another rank / world size: use Data(opt, rank, world)")
            self.net.train()
            self.net.enable_training()
            self.optimizer = make_optimizer(opt, self.net)
            self.scheduler = make_scheduler(opt, self.optimizer)
            if self.dual_model:
                self.dual_models = self.model.dual_models
                self.dual_optimizers = make_dual_optimizer(opt, self.dual_models)
                self.dual_scheduler = make_dual_scheduler(opt, self.dual_optimizers)
            self.reducer = GradReducer().attach(self.net) if self.world > 1 else None
            # one GPU, DRCT, '1*L1': the whole step is one replayed hipGraph per batch shape (opt.train_graph = False keeps the
            # eager launches).  Data-parallel runs stay eager: the bucket hooks launch RCCL all-reduces mid-backward.
            graphed = self.world == 1 and _plain_l1(my_loss) and getattr(opt, "train_graph", True)
            self.graph_step = GraphedTrainStep(self.net, self.optimizer) if graphed and not dual_model else None
            # DRN-L with its dual models, one GPU, '1*L1': the same, with the composite loss and every Adam inside the graph
            self.drn_graph_step = (GraphedDrnTrainStep(self.net, self.dual_models, self.optimizer, self.dual_optimizers,
                                                       getattr(opt, "dual_weight", 0.1))
                                   if graphed and dual_model and all(isinstance(o, TensorAdam) for o in self.dual_optimizers) else None)
        else:
            self.optimizer = self.scheduler = self.reducer = self.graph_step = self.drn_graph_step = None
        self.error_last = 1e8
        self.device = next(self.net.parameters()).device

    def get_last_epoch(self):
        return self.scheduler.last_epoch

    def _log(self, msg, refresh=False):
        if self.rank == 0:
            self.ckp.write_log(msg, refresh) if refresh else self.ckp.write_log(msg)

    def train(self):
        epoch = self.scheduler.last_epoch + 1
        lr = self.scheduler.get_last_lr()[0]
        self._log('[Epoch {}]\tLearning rate: {:.2e}'.format(epoch, Decimal(lr)))
        if self.loss is not None:
            self.loss.start_log()
        self.model.train()
        if hasattr(self.loader_train, "set_epoch"):
            self.loader_train.set_epoch(epoch)
        timer_data, timer_model = timer(), timer()
        n_batches = 0
        for batch, (lr, hr, _) in enumerate(self.loader_train):
            lr, hr = self.prepare(lr, hr)
            timer_data.hold()
            timer_model.tic()
            if self.dual_model and self.drn_graph_step is not None:
                self.drn_graph_step(lr, hr)
                if self.loss is not None:
                    self.loss.note([self.drn_graph_step.logged])     # sum of every term, as the reference's Loss adds them up
            elif self.dual_model:
                drn_train_step(self.net, self.dual_models, lr, hr, self.optimizer, self.dual_optimizers,
                               getattr(self.opt, "dual_weight", 0.1), self.reducer, self.loss)
            elif self.graph_step is not None:
                value = self.graph_step(lr[0], hr)
                if self.loss is not None:
                    self.loss.note([value])
            else:
                train_step(self.net, lr[0], hr, self.optimizer, self.reducer, None if _plain_l1(self.loss) else self.loss,
                           note=self.loss)
            timer_model.hold()
            n_batches = batch + 1
            if (batch + 1) % self.opt.print_every == 0:
                shown = self.loss.display_loss(batch) if self.loss is not None else ''
                self._log('[{}/{}]\t{}\t{:.1f}+{:.1f}s'.format((batch + 1) * self.opt.batch_size, len(self.loader_train.dataset),
                                                                shown, timer_model.release(), timer_data.release()))
            timer_data.tic()
        if self.loss is not None:
            self.loss.end_log(max(n_batches, 1))
            self.error_last = self.loss.log[-1, -1]
        self.step()

    def _validate(self, s: int):
        """One pass over ``loader_test`` at scale ``s``: SR -> round to the u8 grid -> PSNR / SSIM against HR (images that come
        without HR are only super-resolved and saved).  Returns the two sums."""
        psnr_sum = ssim_sum = 0.0
        for lr, hr, names in self.loader_test:
            has_hr = hr.nelement() != 1                              # the loader's placeholder for "no ground truth" is one element
            if has_hr:
                lr, hr = self.prepare(lr, hr)
            else:
                lr, = self.prepare(lr)
            sr = self.model(lr[0])
            sr = quantize(sr[-1] if isinstance(sr, list) else sr, self.opt.rgb_range)
            if has_hr:
                psnr_sum += M.psnr_torch(sr, hr, self.opt.rgb_range)
                ssim_sum += M.ssim_torch(sr, hr, self.opt.rgb_range, win_size=11)
            if self.opt.save_results and self.rank == 0:
                self.ckp.save_results_nopostfix(names[0], sr, s)
        return psnr_sum, ssim_sum

    def test(self):
        """src/trainer.py:242-304: validation PSNR / SSIM of the largest scale into a new row of ``ckp.log``, the reference's
        result line (current value, best value and its epoch), back to train mode."""
        self._log('\nEvaluation:')
        self.ckp.add_log(torch.zeros(1, 2))
        self.model.eval()
        watch = timer()
        n = len(self.loader_test)
        table = self.ckp.log
        with torch.no_grad():
            for si, s in enumerate([max(self.scale)]):
                psnr_sum, ssim_sum = self._validate(s)
                # the reference stores PSNR at column si and SSIM at column 2 si + 1 and reads PSNR back from column 2 si
                # (the same cell for its single scale): kept
                table[-1, si], table[-1, 2 * si + 1] = psnr_sum / n, ssim_sum / n
                top, at = table.max(0)
                self._log('[{} x{}]\tPSNR: {:.2f} (Best: {:.2f} @epoch {})\tSSIM: {:.4f} (Best: {:.4f} @epoch {})'.format(
                    self.opt.data_test, s, table[-1, 2 * si], top[2 * si], at[2 * si] + 1,
                    table[-1, 2 * si + 1], top[2 * si + 1], at[2 * si + 1] + 1))
        self._log('Total time: {:.2f}s\n'.format(watch.toc()), refresh=True)
        self.model.train()

    def step(self):
        """End of an epoch: every cosine schedule advances once (src/trainer.py:312-316)."""
        for sch in [self.scheduler] + (list(self.dual_scheduler) if self.dual_model else []):
            sch.step()

    def prepare(self, *args):
        device = self.model.device if hasattr(self.model, 'device') else self.device
        if len(args) > 1:
            return [a.to(device, non_blocking=True) for a in args[0]], args[-1].to(device, non_blocking=True)
        return [a.to(device, non_blocking=True) for a in args[0]],

    def terminate(self):
        """True when the run is over; a ``test_only`` run validates once on the way out (src/trainer.py:332-340)."""
        if not self.opt.test_only:
            return self.scheduler.last_epoch >= self.opt.epochs
        self.test()
        return True
