"""Training step around the HIP engine (reference src/trainer.py:49-83,152-222): fused Adam on the flat
parameter buffer, the cosine schedule, the data-parallel gradient reducer, and ``train_step``.

Data parallelism (BASELINE config C4): one process per GPU, every rank holds the full model and its own
slice of the minibatch; the only exchange is the gradient all-reduce.  The engine's backward reports each
gradient bucket (one per RDG, in completion order) as soon as its last kernel is enqueued; ``GradReducer``
then all-reduces that contiguous slice of the flat gradient buffer on a second stream while the backward
of the earlier layers is still running.  With the ``nccl`` backend that is RCCL over xGMI.
"""
from __future__ import annotations

import math
import struct
from typing import List, Optional

import torch

from . import _lib as L


class FusedAdam:
    """torch.optim.Adam arithmetic (src/trainer.py:49-59: betas (0.9, 0.999), eps 1e-8, L2 weight decay)
    as ONE kernel over the model's flat parameter buffer (C ABI ``srad_adam_step``)."""

    def __init__(self, model, lr: float = 1e-4, betas=(0.9, 0.999), eps: float = 1e-8, weight_decay: float = 0.0):
        if getattr(model, "flat_params", None) is None:
            model.enable_training()
        self.model = model
        self.lr, self.betas, self.eps, self.weight_decay = float(lr), tuple(betas), float(eps), float(weight_decay)
        self.exp_avg = torch.zeros_like(model.flat_params)
        self.exp_avg_sq = torch.zeros_like(model.flat_params)
        self.step_count = 0
        self.param_groups = [{"lr": self.lr}]          # what lr schedulers read / write

    def zero_grad(self, set_to_none: bool = False) -> None:
        self.model.zero_grad()

    def step(self, grad_scale: float = 1.0) -> None:
        m = self.model
        self.step_count += 1
        lr = float(self.param_groups[0]["lr"])
        L.check(L.lib().srad_adam_step(L.dptr(m.flat_params), L.dptr(m.flat_grads), L.dptr(self.exp_avg),
                                       L.dptr(self.exp_avg_sq), m.flat_params.numel(), lr, self.betas[0], self.betas[1],
                                       self.eps, self.weight_decay, self.step_count, grad_scale,
                                       L.current_stream_ptr()), "adam_step")
        m.mark_params_dirty()

    def hyper(self, grad_scale: float = 1.0) -> List[float]:
        """[lr, 1 - beta1^t, sqrt(1 - beta2^t), grad_scale] of the NEXT step (``srad_adam_step_dev``)."""
        t = self.step_count + 1
        b1, b2 = (struct.unpack("f", struct.pack("f", b))[0] for b in self.betas)     # the betas as the C ABI sees them (fp32)
        return [float(self.param_groups[0]["lr"]), 1.0 - b1 ** t, math.sqrt(1.0 - b2 ** t), float(grad_scale)]

    def step_dev(self, dev_hyper: torch.Tensor) -> None:
        """``step()`` with the per-step scalars read from ``dev_hyper`` (4 floats on the GPU, see ``hyper()``): the launch
        carries no step-dependent argument, so it can sit inside a captured hipGraph.  The caller bumps ``step_count``."""
        m = self.model
        L.check(L.lib().srad_adam_step_dev(L.dptr(m.flat_params), L.dptr(m.flat_grads), L.dptr(self.exp_avg),
                                           L.dptr(self.exp_avg_sq), m.flat_params.numel(), self.betas[0], self.betas[1],
                                           self.eps, self.weight_decay, L.dptr(dev_hyper), L.current_stream_ptr()),
                "adam_step_dev")
        m.mark_params_dirty()

    def state_dict(self):
        return {"step": self.step_count, "exp_avg": self.exp_avg, "exp_avg_sq": self.exp_avg_sq,
                "lr": self.param_groups[0]["lr"]}

    def load_state_dict(self, sd):
        self.step_count = int(sd["step"])
        self.exp_avg.copy_(sd["exp_avg"])
        self.exp_avg_sq.copy_(sd["exp_avg_sq"])
        self.param_groups[0]["lr"] = float(sd["lr"])


class TensorAdam:
    """The same Adam kernel applied tensor by tensor: the optimizer of a dual regression model (two small convolution
    weights; src/trainer.py:62-73).  ``step(grad_scale)`` divides the (all-reduced) gradients by the world size."""

    def __init__(self, params, lr: float = 1e-4, betas=(0.9, 0.999), eps: float = 1e-8, weight_decay: float = 0.0):
        self.params = [p for p in params]
        self.lr, self.betas, self.eps, self.weight_decay = float(lr), tuple(betas), float(eps), float(weight_decay)
        self.state = [(torch.zeros_like(p.data), torch.zeros_like(p.data)) for p in self.params]
        self.step_count = 0
        self.param_groups = [{"lr": self.lr}]

    def zero_grad(self, set_to_none: bool = True) -> None:
        for p in self.params:
            p.grad = None

    def step(self, grad_scale: float = 1.0) -> None:
        self.step_count += 1
        lr = float(self.param_groups[0]["lr"])
        for p, (m, v) in zip(self.params, self.state):
            if p.grad is None:
                continue
            g = p.grad.detach().float().contiguous()
            if not p.data.is_contiguous():
                raise RuntimeError("TensorAdam needs contiguous parameters")
            L.check(L.lib().srad_adam_step(L.dptr(p.data), L.dptr(g), L.dptr(m), L.dptr(v), p.numel(), lr, self.betas[0],
                                           self.betas[1], self.eps, self.weight_decay, self.step_count, float(grad_scale),
                                           L.current_stream_ptr()), "adam_step")

    def hyper(self, grad_scale: float = 1.0) -> List[float]:
        """[lr, 1 - beta1^t, sqrt(1 - beta2^t), grad_scale] of the NEXT step (as ``FusedAdam.hyper``)."""
        t = self.step_count + 1
        b1, b2 = (struct.unpack("f", struct.pack("f", b))[0] for b in self.betas)
        return [float(self.param_groups[0]["lr"]), 1.0 - b1 ** t, math.sqrt(1.0 - b2 ** t), float(grad_scale)]

    def step_dev(self, dev_hyper: torch.Tensor) -> None:
        """``step()`` with the per-step scalars on the GPU (inside a captured hipGraph); the caller bumps ``step_count``."""
        for p, (m, v) in zip(self.params, self.state):
            if p.grad is None:
                continue
            g = p.grad.detach().float().contiguous()
            if not p.data.is_contiguous():
                raise RuntimeError("TensorAdam needs contiguous parameters")
            L.check(L.lib().srad_adam_step_dev(L.dptr(p.data), L.dptr(g), L.dptr(m), L.dptr(v), p.numel(), self.betas[0],
                                               self.betas[1], self.eps, self.weight_decay, L.dptr(dev_hyper),
                                               L.current_stream_ptr()), "adam_step_dev")

    def state_dict(self):
        return {"step": self.step_count, "lr": self.param_groups[0]["lr"],
                "exp_avg": [m for m, _ in self.state], "exp_avg_sq": [v for _, v in self.state]}

    def load_state_dict(self, sd):
        self.step_count = int(sd["step"])
        self.param_groups[0]["lr"] = float(sd["lr"])
        for (m, v), a, b in zip(self.state, sd["exp_avg"], sd["exp_avg_sq"]):
            m.copy_(a)
            v.copy_(b)


def cosine_lr(base_lr: float, epoch: int, t_max: float, eta_min: float) -> float:
    """CosineAnnealingLR closed form (src/trainer.py:76-83: T_max = epochs, stepped once per epoch)."""
    return eta_min + (base_lr - eta_min) * (1.0 + math.cos(math.pi * epoch / t_max)) / 2.0


def bucket_slices(buckets, world_size: int):
    """[(offset, length)] of the flat gradient buffer per bucket - host logic shared with the CPU tests."""
    return [(int(a), int(n)) for a, n in buckets if n > 0]


class GradReducer:
    """Bucketed gradient all-reduce overlapped with the backward.  ``attach(model)`` installs the engine's
    bucket hook; ``finish()`` (after backward) makes the compute stream wait for the reductions.  Gradients
    are SUMMED; the 1/world_size goes into the optimizer's ``grad_scale``."""

    def __init__(self, process_group=None, overlap: bool = True):
        import torch.distributed as dist
        self.dist = dist
        self.group = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        self.overlap = overlap
        self.comm_stream = None
        self.model = None
        self._pending = []

    def attach(self, model) -> "GradReducer":
        self.model = model
        if self.world > 1:
            model.on_bucket = self._on_bucket
        return self

    def _on_bucket(self, b: int) -> None:
        off, n = self.model.grad_buckets[b]
        if n <= 0:
            return
        g = self.model.flat_grads[off:off + n]
        if g.is_cuda and self.overlap:
            if self.comm_stream is None:
                self.comm_stream = torch.cuda.Stream(device=g.device)
            self.comm_stream.wait_stream(torch.cuda.current_stream())     # bucket b is final on the compute stream
            with torch.cuda.stream(self.comm_stream):
                self.dist.all_reduce(g, group=self.group)
        else:
            self.dist.all_reduce(g, group=self.group)

    def reduce_all(self, flat_grads: torch.Tensor, buckets) -> None:
        """Non-overlapped form (also what the gloo CPU test drives): reduce every bucket in order."""
        if self.world == 1:
            return
        for off, n in bucket_slices(buckets, self.world):
            self.dist.all_reduce(flat_grads[off:off + n], group=self.group)

    def finish(self) -> None:
        if self.comm_stream is not None:
            torch.cuda.current_stream().wait_stream(self.comm_stream)

    @property
    def grad_scale(self) -> float:
        return 1.0 / self.world


def train_step(model, lr_img: torch.Tensor, hr_img: torch.Tensor, optimizer: FusedAdam,
               reducer: Optional[GradReducer] = None, loss_fn=None, note=None) -> torch.Tensor:
    """One iteration of Trainer.train for DRCT (src/trainer.py:161-205): zero_grad, forward, loss, backward,
    (all-reduce,) Adam.  Returns the loss as a 0-d device tensor (no host sync).  ``loss_fn``: a ``loss.Loss`` whose
    ``value_and_grad`` replaces the built-in nn.L1Loss; ``note``: a ``loss.Loss`` that only logs the value."""
    from . import metrics as M
    optimizer.zero_grad()
    sr = model._forward_train(lr_img)
    if loss_fn is not None:
        loss, dy = loss_fn.value_and_grad(sr, hr_img)
    else:
        loss = M.l1_loss(sr, hr_img)
        dy = torch.empty_like(sr)
        L.check(L.lib().srad_l1_grad(L.dptr(sr), L.dptr(hr_img), L.dptr(dy), sr.numel(), 1.0 / sr.numel(),
                                     L.current_stream_ptr()), "l1_grad")
        if note is not None:
            note.note([loss])
    model._backward(dy, need_dx=False)
    if reducer is not None:
        reducer.finish()
    optimizer.step(grad_scale=reducer.grad_scale if reducer is not None else 1.0)
    return loss


class GraphedTrainStep:
    """``train_step`` for one GPU as ONE hipGraph: the ~700 launches of a DRCT training step (re-pack, zero_grad, forward,
    L1, the two-stream backward, Adam) are captured once per batch shape and replayed, which removes the 3-5 us
    dispatch gap between dependent launches.  Per step only the two input copies, a 16-byte copy of Adam's scalars
    (``FusedAdam.hyper``) and the graph launch are issued.  DropPath masks come from ``torch.rand`` inside the capture
    (PyTorch's graph-safe generator advances the Philox offset per replay).  With a ``GradReducer`` (world > 1) the
    bucket hooks launch collectives from the host in the middle of the backward, so ``train_step`` runs eagerly."""

    def __init__(self, model, optimizer: FusedAdam, warmup: int = 2):
        self.model, self.optimizer, self.warmup = model, optimizer, int(warmup)
        self._graphs = {}          # (lr shape, hr shape) -> (graph, static lr, static hr, loss, hyper)
        self._eager_calls = {}

    def _body(self, lr_img, hr_img, hyper):
        from . import metrics as M
        m, opt = self.model, self.optimizer
        opt.zero_grad()
        sr = m._forward_train(lr_img)
        loss = M.l1_loss(sr, hr_img)
        dy = torch.empty_like(sr)
        L.check(L.lib().srad_l1_grad(L.dptr(sr), L.dptr(hr_img), L.dptr(dy), sr.numel(), 1.0 / sr.numel(),
                                     L.current_stream_ptr()), "l1_grad")
        m._backward(dy, need_dx=False)
        opt.step_dev(hyper)
        return loss

    def __call__(self, lr_img: torch.Tensor, hr_img: torch.Tensor) -> torch.Tensor:
        m, opt = self.model, self.optimizer
        key = (tuple(lr_img.shape), tuple(hr_img.shape))
        entry = self._graphs.get(key)
        if entry is None:
            n = self._eager_calls.get(key, 0)
            if n < self.warmup or m.on_bucket is not None or m.keep_scale_override is not None:
                self._eager_calls[key] = n + 1          # allocates workspaces, configures kernels, creates the side stream
                return train_step(m, lr_img, hr_img, opt)
            s_lr, s_hr = lr_img.detach().float().contiguous().clone(), hr_img.detach().float().contiguous().clone()
            hyper = torch.zeros(4, dtype=torch.float32, device=lr_img.device)
            m.mark_params_dirty()                      # the capture starts with the re-pack of the parameters
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, capture_error_mode="thread_local"):
                loss = self._body(s_lr, s_hr, hyper)
            entry = (g, s_lr, s_hr, loss, hyper)
            self._graphs[key] = entry
        g, s_lr, s_hr, loss, hyper = entry
        s_lr.copy_(lr_img, non_blocking=True)
        s_hr.copy_(hr_img, non_blocking=True)
        h = opt.hyper()                                # by value through a one-thread kernel: no host copy, no sync
        L.check(L.lib().srad_set4(L.dptr(hyper), h[0], h[1], h[2], h[3], L.current_stream_ptr()), "set4")
        g.replay()
        opt.step_count += 1
        m.mark_params_dirty()                          # the replayed Adam changed the flat parameters
        return loss.clone()                            # the graph's own output buffer is overwritten by the next replay


def drn_loss(sr, lr_list, hr, sr2lr, dual_weight: float = 0.1, loss_fn=None, terms: bool = False):
    """DRN training loss (src/trainer.py:168-185): loss(sr[-1], hr) + sum_j loss(sr[j], lr[j]) over the coarser outputs
    + dual_weight * sum_i loss(dual_i(sr[i - n]), lr[i]).  ``lr_list`` = [LR_x(max), ..., LR_x2] coarse -> fine, as the
    reference loader yields it.  ``loss_fn``: a ``loss.Loss`` (default: the engine's nn.L1Loss reduction); either way the
    values and the gradients into the SR outputs are the engine's reductions, autograd only chains them.
    ``terms=True`` also returns ``loss_primary + loss_dual`` - the sum of every term as the reference's logging ``Loss`` adds them
    up call by call (the dual terms enter its log unweighted)."""
    if loss_fn is None:
        from .loss import L1Loss
        loss_fn = L1Loss()
    loss_primary = loss_fn(sr[-1], hr)
    for i in range(1, len(sr)):
        loss_primary = loss_primary + loss_fn(sr[i - 1 - len(sr)], lr_list[i - len(sr)])
    loss_dual = loss_fn(sr2lr[0], lr_list[0])
    for i in range(1, len(sr2lr)):
        loss_dual = loss_dual + loss_fn(sr2lr[i], lr_list[i])
    total = loss_primary + dual_weight * loss_dual
    return (total, (loss_primary + loss_dual).detach()) if terms else total


def drn_train_step(model, dual_models, lr_list, hr, optimizer, dual_optimizers, dual_weight: float = 0.1,
                   reducer: Optional["GradReducer"] = None, loss_fn=None, log=None) -> torch.Tensor:
    """One iteration of Trainer.train for DRN-L with its dual regression models (src/trainer.py:161-205): forward of
    the SR net and of every dual model on the matching SR output, the composite loss, backward through the dual models
    into the SR outputs and through the DRN engine, one Adam step for the SR net and one per dual model."""
    optimizer.zero_grad()
    for o in dual_optimizers:
        o.zero_grad()
    sr = model(lr_list[0])
    sr2lr = [dual_models[i](sr[i - len(dual_models)]) for i in range(len(dual_models))]
    loss, logged = drn_loss(sr, lr_list, hr, sr2lr, dual_weight, loss_fn, terms=True)
    if log is not None:
        log.logged = logged                            # (GraphedDrnTrainStep's eager warm-up steps)
    # data parallel: the SR net's buckets (one per level, ``model.grad_buckets``) are all-reduced from the engine's hook while
    # the backward continues (``GradReducer.attach``); a model without buckets is reduced in one piece afterwards
    bucketed = reducer is not None and reducer.world > 1 and bool(getattr(model, "grad_buckets", None))
    if bucketed and reducer.model is not model:
        reducer.attach(model)
    loss.backward()
    scale = 1.0
    if reducer is not None:
        if bucketed:
            reducer.finish()
        else:
            reducer.reduce_all(model.flat_grads, [(0, model.flat_grads.numel())])
        scale = reducer.grad_scale
        for dm, o in zip(dual_models, dual_optimizers):
            for p in dm.parameters():
                reducer.dist.all_reduce(p.grad, group=reducer.group)
                if not isinstance(o, TensorAdam):
                    p.grad.mul_(scale)               # an optimizer without a grad_scale argument sees the mean itself
    optimizer.step(grad_scale=scale)
    for o in dual_optimizers:
        o.step(scale) if isinstance(o, TensorAdam) else o.step()
    return loss.detach()


class GraphedDrnTrainStep:
    """``drn_train_step`` for one GPU as ONE hipGraph (as ``GraphedTrainStep`` for DRCT): the ~1000 launches of a DRN-L step
    with its dual models - zero_grad, re-pack, forward, the composite loss, autograd's chain through the dual models into the
    engine's two-stream backward, Adam for the SR net and for every dual model - are captured once per batch shape and
    replayed.  Per step only the input copies, one 16-byte copy of Adam's scalars per optimizer and the graph launch are
    issued.  The dual optimizers must be ``TensorAdam`` (the engine's kernel, scalars on the GPU); with a ``GradReducer`` the
    step runs eagerly (collectives are launched from the host inside the backward)."""

    def __init__(self, model, dual_models, optimizer: FusedAdam, dual_optimizers, dual_weight: float = 0.1, loss_fn=None,
                 warmup: int = 2, reducer: Optional["GradReducer"] = None):
        self.reducer = reducer if reducer is not None and reducer.world > 1 else None     # data parallel: every step runs eagerly
        self.logged = None                             # after a call: the value the reference's Loss log adds up for the step
        for o in dual_optimizers:
            if not isinstance(o, TensorAdam):
                raise TypeError("GraphedDrnTrainStep needs TensorAdam dual optimizers (torch.optim steps read host scalars)")
        if hasattr(loss_fn, "note"):
            # a loss.Loss keeps its log by calling note() from Python on every forward: inside a capture that runs once, and the
            # replays would re-add the capture-time values.  The Trainer logs through the eager step.
            raise TypeError("GraphedDrnTrainStep takes loss_fn=None (the reference's 1*L1) or a plain callable, not a logging loss.Loss; "
                            "use drn_train_step for that")
        self.model, self.duals, self.optimizer, self.dual_optimizers = model, list(dual_models), optimizer, list(dual_optimizers)
        self.dual_weight, self.loss_fn, self.warmup = float(dual_weight), loss_fn, int(warmup)
        self._graphs = {}
        self._eager_calls = {}

    def _body(self, lrs, hr, hypers):
        self.optimizer.zero_grad()
        for o in self.dual_optimizers:
            o.zero_grad()
        sr = self.model(lrs[0])
        sr2lr = [self.duals[i](sr[i - len(self.duals)]) for i in range(len(self.duals))]
        loss, logged = drn_loss(sr, lrs, hr, sr2lr, self.dual_weight, self.loss_fn, terms=True)
        loss.backward()
        self.optimizer.step_dev(hypers[0])
        for o, h in zip(self.dual_optimizers, hypers[1:]):
            o.step_dev(h)
        return loss.detach(), logged

    def __call__(self, lr_list, hr: torch.Tensor) -> torch.Tensor:
        m = self.model
        key = (tuple(tuple(t.shape) for t in lr_list), tuple(hr.shape))
        entry = self._graphs.get(key)
        if entry is None:
            n = self._eager_calls.get(key, 0)
            if n < self.warmup or self.reducer is not None or getattr(m, "on_bucket", None) is not None:   # bucket hooks launch collectives from the host: eager
                self._eager_calls[key] = n + 1          # allocates workspaces, configures kernels, creates the side stream
                value = drn_train_step(m, self.duals, lr_list, hr, self.optimizer, self.dual_optimizers, self.dual_weight,
                                       self.reducer, self.loss_fn, log=self)
                return value
            s_lrs = [t.detach().float().contiguous().clone() for t in lr_list]
            s_hr = hr.detach().float().contiguous().clone()
            hypers = [torch.zeros(4, dtype=torch.float32, device=hr.device) for _ in range(1 + len(self.dual_optimizers))]
            m.mark_params_dirty()                      # the capture starts with the re-pack of the parameters
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                loss, logged = self._body(s_lrs, s_hr, hypers)
            entry = (g, s_lrs, s_hr, loss, hypers, logged)
            self._graphs[key] = entry
        g, s_lrs, s_hr, loss, hypers, logged = entry
        for a, b in zip(s_lrs, lr_list):
            a.copy_(b, non_blocking=True)
        s_hr.copy_(hr, non_blocking=True)
        for o, h in zip([self.optimizer] + self.dual_optimizers, hypers):
            v = o.hyper()
            L.check(L.lib().srad_set4(L.dptr(h), v[0], v[1], v[2], v[3], L.current_stream_ptr()), "set4")
        g.replay()
        for o in [self.optimizer] + self.dual_optimizers:
            o.step_count += 1
        m.mark_params_dirty()                          # the replayed Adam changed the flat parameters
        self.logged = logged.clone()                   # what the reference's Loss log adds up for this step (drn_loss, terms=True)
        return loss.clone()

