"""Host-side mirrors of the reference networks: ``DRCT`` (reference src/drct.py:716-898) and
``DRN`` (src/drn.py:160-270).  They are ``torch.nn.Module``s only for what PyTorch is used for
here - owning device memory for parameters under the reference's state-dict names, streams, and
``torch.distributed``; ``forward`` enqueues the hand-written HIP engine through the C ABI
(include/srad.h).  There is no eager / CPU fallback.
"""
from __future__ import annotations

import ctypes as C
import math
from typing import Dict, List, Optional

import numpy as np
import torch
import torch.nn as nn

from . import _lib as L
from . import spec as S


# ------------------------------------------------------------------ parameter tree
class _Node(nn.Module):
    """Plain container so that parameters live under dotted reference names."""


def _attach(root: nn.Module, dotted: str, tensor: torch.Tensor, is_param: bool) -> None:
    parts = dotted.split(".")
    mod = root
    for p in parts[:-1]:
        if p not in mod._modules:
            mod.add_module(p, _Node())
        mod = mod._modules[p]
    if is_param:
        mod.register_parameter(parts[-1], nn.Parameter(tensor))
    else:
        mod.register_buffer(parts[-1], tensor)


def _trunc_normal_(t: torch.Tensor, std: float, a: float = -2.0, b: float = 2.0) -> None:
    """Same distribution as the reference's trunc_normal_ (src/drct.py:32-93)."""
    nn.init.trunc_normal_(t, mean=0.0, std=std, a=a, b=b)


def _reference_init(name: str, kind: str, shape, cfg) -> torch.Tensor:
    """Initial values with the reference's distributions: Linear weights
    trunc-N(0,.02), Linear bias 0, LayerNorm (1,0), bias table trunc-N(.02); convolutions keep
    PyTorch's default (kaiming-uniform(a=sqrt 5) weight, U(+-1/sqrt(fan_in)) bias)."""
    if kind in ("index", "mask", "meanshift_w", "meanshift_b_sub", "meanshift_b_add"):
        a = S.synth_tensor(name, shape, kind, cfg=cfg)
        if a.flags.writeable or a.nbytes < (1 << 24):
            return torch.from_numpy(np.array(a))
        # the large index / mask buffers of wide windows (134 MB / 1 GB each at window 64, identical in every block):
        # all blocks alias one read-only host array until the module is moved to its GPU
        import warnings
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            return torch.from_numpy(a)
    t = torch.empty(shape, dtype=torch.float32)
    if kind == "lin_w":
        _trunc_normal_(t, 0.02)
    elif kind == "lin_b" or kind == "ln_b":
        t.zero_()
    elif kind == "ln_w":
        t.fill_(1.0)
    elif kind == "table":
        _trunc_normal_(t, 0.02)
    elif kind == "conv_w":
        nn.init.kaiming_uniform_(t, a=math.sqrt(5))
    elif kind == "conv_b":
        # fan_in is not recoverable from the bias alone; callers patch conv biases afterwards
        t.zero_()
    else:
        raise ValueError(kind)
    return t


class _EngineModule(nn.Module):
    """Shared plumbing: parameter tree, weight arena, change tracking, workspace cache."""

    _prefix = ""            # C symbol prefix: srad_drct / srad_drn

    def _init_engine(self, spec, cfg, precision: str, use_graph: bool):
        self._spec = spec
        self._cfg = cfg
        self.precision = precision
        self.use_graph = bool(use_graph)
        self._handle = None
        self._arena = None
        self._param_cache = None
        self._ws: Dict[tuple, torch.Tensor] = {}
        self._static_io: Dict[tuple, tuple] = {}
        self._side_stream = None
        conv_fan = {}
        for name, (shape, kind) in spec.items():
            t = _reference_init(name, kind, shape, cfg)
            if kind == "conv_w":
                conv_fan[name[:-len("weight")]] = int(np.prod(shape[1:]))
            _attach(self, name, t, is_param=kind not in ("index", "mask"))
        with torch.no_grad():
            for name, (shape, kind) in spec.items():
                if kind == "conv_b":
                    bound = 1.0 / math.sqrt(conv_fan[name[:-len("bias")]])
                    self.get_parameter(name).uniform_(-bound, bound)

    # -- C handle -------------------------------------------------------------------------
    def _fn(self, suffix):
        return getattr(L.lib(), f"{self._prefix}_{suffix}")

    def _make_handle(self):
        raise NotImplementedError

    def _ensure_handle(self, device):
        if self._handle is not None:
            return
        self._handle = self._make_handle()
        nbytes = C.c_size_t()
        L.check(self._fn("arena_bytes")(self._handle, C.byref(nbytes)), "arena_bytes")
        self._arena = torch.zeros(nbytes.value + 256, dtype=torch.uint8, device=device)
        off = (-self._arena.data_ptr()) % 256
        self._arena_ptr = self._arena.data_ptr() + off
        L.check(self._fn("bind_arena")(self._handle, C.c_void_p(self._arena_ptr), C.c_size_t(nbytes.value)), "bind_arena")
        n = self._fn("num_params")(self._handle)
        self._engine_params = []
        name = C.c_char_p()
        numel = C.c_int64()
        for i in range(n):
            L.check(self._fn("param_info")(self._handle, i, C.byref(name), C.byref(numel)), "param_info")
            self._engine_params.append((name.value.decode(), numel.value))

    def __del__(self):
        try:
            if getattr(self, "_handle", None) is not None:
                self._fn("destroy")(self._handle)
                self._handle = None
        except Exception:
            pass

    def _sync_weights(self, device):
        """(Re)pack every parameter whose storage or version changed since the last forward
        (optimizer steps and load_state_dict bump ``_version``).  The steady-state cost is one pass
        over a cached tensor list (no module-tree walks on the hot path)."""
        self._ensure_handle(device)
        if getattr(self, "flat_params", None) is not None:
            # training mode was enabled: the parameters live in the flat buffer and ONE launch re-packs all of them
            # (forward packs included), whoever changed them - the fused Adam, a torch optimizer or load_state_dict
            self._sync_flat()
            return
        if getattr(self, "_param_cache", None) is None:
            self._param_cache = [(name, self.get_parameter(name)) for name, _ in self._engine_params]
            self._tags = [None] * len(self._param_cache)
        tags = [(p.data_ptr(), p._version) for _, p in self._param_cache]
        if tags == self._tags:
            return
        stream = L.current_stream_ptr()
        for i, (name, p) in enumerate(self._param_cache):
            if tags[i] == self._tags[i]:
                continue
            if p.device != device or p.dtype != torch.float32:
                raise RuntimeError(f"{name}: parameters must be fp32 on {device} (got {p.dtype} on {p.device})")
            src = p.detach()
            if not src.is_contiguous():
                src = src.contiguous()
            L.check(self._fn("set_param")(self._handle, name.encode(), L.dptr(src), C.c_int64(src.numel()), stream),
                    f"set_param({name})")
        self._tags = tags

    def _apply(self, fn, *a, **kw):
        # .to()/.cuda()/.float() replace parameter storage: drop the cached tensor list
        self._param_cache = None
        return super()._apply(fn, *a, **kw)

    def _workspace(self, B, H, W, device) -> torch.Tensor:
        key = (B, H, W)
        ws = self._ws.get(key)
        if ws is None:
            nbytes = C.c_size_t()
            L.check(self._fn("workspace_bytes")(self._handle, B, H, W, C.byref(nbytes)), "workspace_bytes")
            ws = torch.empty(nbytes.value + 256, dtype=torch.uint8, device=device)
            self._ws = {key: ws}          # keep one shape resident
        return ws

    def _run(self, fn):
        """Run ``fn()`` (which enqueues on the current stream).  With graph replay enabled the work
        goes to a private non-default stream (the legacy default stream cannot be captured),
        ordered after / before the caller's stream with events."""
        if not self.use_graph:
            return fn()
        cur = torch.cuda.current_stream()
        if self._side_stream is None:
            self._side_stream = torch.cuda.Stream(device=cur.device)
        self._side_stream.wait_stream(cur)
        with torch.cuda.stream(self._side_stream):
            out = fn()
        cur.wait_stream(self._side_stream)
        for t in (out if isinstance(out, (list, tuple)) else [out]):
            t.record_stream(cur)           # allocated on the side stream, consumed on the caller's
        return out

    @staticmethod
    def _aligned(t: torch.Tensor):
        off = (-t.data_ptr()) % 256
        return C.c_void_p(t.data_ptr() + off), C.c_size_t(t.numel() - off)

    def _check_input(self, x: torch.Tensor, channels: int):
        if not isinstance(x, torch.Tensor) or x.dim() != 4:
            raise ValueError("expected a [B, C, H, W] tensor")
        if x.shape[1] != channels:
            raise ValueError(f"expected {channels} channels, got {x.shape[1]}")
        if not x.is_cuda:
            raise RuntimeError("srad_amd runs on the GPU only (HIP engine); got a CPU tensor - there is no CPU fallback")
        if self.training and torch.is_grad_enabled() and not self._can_train():
            raise NotImplementedError(f"{type(self).__name__}: the HIP backward pass is not built for this model: "
                                      "call under torch.no_grad() / .eval()")
        return x.detach().to(torch.float32).contiguous()

    def _can_train(self) -> bool:
        return False

    # -- training plumbing shared by DRCT and DRN (C ABI <prefix>_train_*, <prefix>_sync_params) -----------
    def enable_training(self):
        """Re-home every parameter into ONE flat fp32 device buffer (``p.data`` become views, state-dict and
        optimizers keep working) with a twin flat gradient buffer (``p.grad`` views), and bind the engine's
        training arena.  Call after the module is on its GPU; moving it afterwards needs another call."""
        dev = next(self.parameters()).device
        if dev.type != "cuda":
            raise RuntimeError("srad_amd trains on the GPU only (HIP engine); move the model with .cuda() first")
        if L.PRECISIONS[self.precision] == L.PREC_BF16X3:
            raise RuntimeError("the split-bf16 precision ('bf16x3') is an inference mode; train with precision 'fp32' or 'bf16'")
        self._ensure_handle(dev)
        h = self._handle
        total = C.c_int64()
        L.check(self._fn("train_param_floats")(h, C.byref(total)), "train_param_floats")
        flat = torch.zeros(total.value, dtype=torch.float32, device=dev)
        grad = torch.zeros(total.value, dtype=torch.float32, device=dev)
        views, homes = [], []
        off = C.c_int64()
        with torch.no_grad():
            for i, (name, numel) in enumerate(self._engine_params):
                L.check(self._fn("train_param_offset")(h, i, C.byref(off)), "train_param_offset")
                p = self.get_parameter(name)
                v = flat[off.value:off.value + numel].view(p.shape)
                v.copy_(p.data)
                p.data = v
                g = grad[off.value:off.value + numel].view(p.shape)
                p.grad = g
                views.append((p, g))
                homes.append((p, v.data_ptr(), v))
        nbytes = C.c_size_t()
        L.check(self._fn("train_arena_bytes")(h, C.byref(nbytes)), "train_arena_bytes")
        self._tarena = torch.zeros(nbytes.value + 256, dtype=torch.uint8, device=dev)
        tp, tb = self._aligned(self._tarena)
        L.check(self._fn("train_bind")(h, tp, tb), "train_bind")
        self.flat_params, self.flat_grads, self._grad_views = flat, grad, views
        self._flat_homes = homes
        self._synced_version = None
        self._train_ws = {}
        self._anchor = torch.zeros(1, device=dev, requires_grad=True)
        self._param_cache = None
        return self

    def mark_params_dirty(self) -> None:
        """Tell the engine the flat parameters were changed by something that does not bump tensor versions
        (the fused Adam kernel, a replayed hipGraph)."""
        self._synced_version = None
        self._tags = None

    def _flat_state(self):
        """Everything that can change the packed weights behind the engine's back.  After ``enable_training`` every
        Parameter is a view into ``flat_params`` but keeps its OWN version counter: ``torch.optim.Adam.step()`` and
        ``load_state_dict`` bump ``p._version`` and leave ``flat_params._version`` alone.  A parameter whose storage
        was replaced (``p.data = t``, ``load_state_dict(assign=True)``) is copied back into its slot and re-homed."""
        ver = self.flat_params._version
        for p, ptr, home in self._flat_homes:
            ver += p._version
            if p.data_ptr() != ptr:
                with torch.no_grad():
                    home.copy_(p.data)
                    p.data = home
                ver += 1 << 40
        return ver

    def _sync_flat(self) -> None:
        if getattr(self, "flat_params", None) is None:
            raise RuntimeError(f"{type(self).__name__}: call enable_training() before a training-mode forward")
        ver = self._flat_state()
        if self._synced_version != ver:
            L.check(self._fn("sync_params")(self._handle, L.dptr(self.flat_params), L.current_stream_ptr()), "sync_params")
            self._synced_version = self._flat_state()      # re-homing above does not repeat
            self._tags = None

    def _train_workspace(self, B, H, W, dev):
        key = (B, H, W)
        if key not in self._train_ws:
            nbytes = C.c_size_t()
            L.check(self._fn("train_workspace_bytes")(self._handle, B, H, W, C.byref(nbytes)), "train_workspace_bytes")
            self._train_ws = {key: torch.empty(nbytes.value + 256, dtype=torch.uint8, device=dev)}
        return self._aligned(self._train_ws[key])

    def zero_grad(self, set_to_none: bool = False) -> None:   # noqa: D401 - nn.Module signature
        if getattr(self, "flat_grads", None) is not None:
            self.flat_grads.zero_()
            if self._grad_views[0][0].grad is None or self._grad_views[-1][0].grad is None:
                for p, g in self._grad_views:       # an optimizer dropped the views (set_to_none): re-attach all 900
                    p.grad = g
        else:
            super().zero_grad(set_to_none)

    def flops(self, B: int, H: int, W: int) -> float:
        dev = next(self.parameters()).device
        self._ensure_handle(dev)
        f = C.c_double()
        L.check(self._fn("flops")(self._handle, B, H, W, C.byref(f)), "flops")
        return f.value


class _DrctTrainFn(torch.autograd.Function):
    """autograd seam of the training step: forward = srad_drct_forward_train, backward = srad_drct_backward.
    Parameter gradients are written straight into the module's flat gradient buffer (``p.grad`` views);
    only dLoss/dx travels through autograd.  One forward may be outstanding per module."""

    @staticmethod
    def forward(ctx, x, anchor, module):
        ctx.module = module
        ctx.need_dx = x.requires_grad
        return module._forward_train(x.detach().to(torch.float32).contiguous())

    @staticmethod
    def backward(ctx, dy):
        dx = ctx.module._backward(dy.to(torch.float32).contiguous(), ctx.need_dx)
        return dx, None, None


class _DrnTrainFn(torch.autograd.Function):
    """autograd seam of the DRN training step: the outputs are the phase + 1 images; backward hands the engine one
    gradient per output (None for outputs the loss does not use).  The LR input gets no gradient (the reference
    never asks for one)."""

    @staticmethod
    def forward(ctx, x, anchor, module):
        ctx.module = module
        outs = module._forward_train(x.detach().to(torch.float32).contiguous())
        return tuple(outs)

    @staticmethod
    def backward(ctx, *dys):
        ctx.module._backward(list(dys))
        return None, None, None


class _DualFn(torch.autograd.Function):
    """Dual regression model with gradients: y = conv(lrelu(conv_s2(x, w0)), w1) (src/model.py:8-44)."""

    @staticmethod
    def forward(ctx, x, w0, w1, module):
        ctx.module = module
        x = x.detach().to(torch.float32).contiguous()
        ctx.save_for_backward(x, w0.detach(), w1.detach())
        return module._run_forward(x, w0.detach().contiguous(), w1.detach().contiguous())

    @staticmethod
    def backward(ctx, dy):
        x, w0, w1 = ctx.saved_tensors
        m = ctx.module
        B, Cc, H, W = x.shape
        dy = dy.to(torch.float32).contiguous()
        dx = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        dw0, dw1 = torch.zeros_like(w0), torch.zeros_like(w1)
        nbytes = C.c_size_t()
        L.check(L.lib().srad_dual_backward_workspace_bytes(B, Cc, H, W, m.n_feats, C.byref(nbytes)), "dual_backward_workspace_bytes")
        ws = torch.empty(nbytes.value + 256, dtype=torch.uint8, device=x.device)
        wp, wb = _EngineModule._aligned(ws)
        L.check(L.lib().srad_dual_backward(L.dptr(w0.contiguous()), L.dptr(w1.contiguous()), Cc, m.n_feats, m.negval, L.dptr(x), B, H, W,
                                           L.dptr(dy), L.dptr(dx), L.dptr(dw0), L.dptr(dw1), wp, wb, L.PRECISIONS[m.precision],
                                           L.current_stream_ptr()), "dual_backward")
        return dx, dw0, dw1, None


# ------------------------------------------------------------------ DRCT
class DRCT(_EngineModule):
    """Drop-in for reference ``src.drct.DRCT(opt)``: same ``opt`` fields, same state-dict, same
    ``forward(x)`` contract (fp32 NCHW in [0, rgb_range] -> [B, C, H*s, W*s])."""

    _prefix = "srad_drct"

    def __init__(self, opt, precision: Optional[str] = None, use_graph: Optional[bool] = None):
        super().__init__()
        depths = tuple(getattr(opt, "depths", (6,) * 12))
        heads = getattr(opt, "num_heads", 6)
        heads = heads[0] if isinstance(heads, (tuple, list)) else heads
        cfg = S.DRCTConfig(in_chans=opt.n_colors, img_size=opt.img_size, window_size=opt.window_size,
                           upscale=opt.upscale, embed_dim=getattr(opt, "embed_dim", 180), n_rdg=len(depths),
                           num_heads=heads, mlp_ratio=float(getattr(opt, "mlp_ratio", 2)),
                           img_range=float(getattr(opt, "img_range", 1.0)),
                           depth_per_rdg=depths[0] if depths else 6)
        if getattr(opt, "upsampler", "pixelshuffle") != "pixelshuffle":
            raise ValueError("only upsampler='pixelshuffle' is supported (the reference CLI never selects another)")
        if getattr(opt, "resi_connection", "1conv") != "1conv":
            raise ValueError("only resi_connection='1conv' is supported")
        self.cfg = cfg
        self.window_size = cfg.window_size
        self.upscale = cfg.upscale
        self.img_range = cfg.img_range
        self.drop_path_rate = float(getattr(opt, "drop_path_rate", 0.1))      # src/drct.py:736
        prec = precision or getattr(opt, "precision", "fp32")
        graph = getattr(opt, "use_graph", True) if use_graph is None else use_graph
        self._init_engine(S.drct_spec(cfg), cfg, prec, graph)

    def _make_handle(self):
        c = self.cfg
        cc = L.DrctConfig(c.in_chans, c.img_size, c.window_size, c.upscale, c.embed_dim, c.n_rdg, c.num_heads, c.gc,
                          c.num_feat, c.mlp_ratio, c.img_range, L.PRECISIONS[self.precision], int(self.use_graph))
        h = C.c_void_p()
        L.check(L.lib().srad_drct_create(C.byref(cc), C.byref(h)), "drct_create")
        return h

    # -- training (C ABI srad_drct_forward_train / srad_drct_backward) ------------------------
    def _can_train(self) -> bool:
        # every window size the reference's CLI builds (window_size = img_size // 4 in {2, 4, 8, 16}, src/main.py:218-219,286):
        # 8 x 8 windows (C2, C4) take the fused kernels, the others the unfused launches and the general attention backward
        return 1 <= self.window_size <= 16

    def enable_training(self) -> "DRCT":
        super().enable_training()
        h = self._handle
        self.keep_scale_override = None     # tests: explicit DropPath factors [2 * n_blocks, B]
        self.on_bucket = None               # data-parallel hook: callable(bucket index), see GradReducer
        nb = L.lib().srad_drct_num_buckets(h)
        a, b = C.c_int64(), C.c_int64()
        self.grad_buckets = []
        for i in range(nb):
            L.check(L.lib().srad_drct_bucket_range(h, i, C.byref(a), C.byref(b)), "bucket_range")
            self.grad_buckets.append((a.value, b.value))
        return self

    def drop_path_keep_probs(self) -> torch.Tensor:
        """keep_prob of every Swin block: all five blocks of RDG i use dpr[i * depth] of
        linspace(0, rate, sum(depths)) (src/drct.py:819,829,332)."""
        c = self.cfg
        rates = S.DRCTConfig.drop_path_probs(type("R", (), dict(n_rdg=c.n_rdg, depth_per_rdg=c.depth_per_rdg,
                                                                 drop_path_rate=self.drop_path_rate))())
        return (1.0 - torch.tensor(rates, dtype=torch.float32)).repeat_interleave(5)

    def _forward_train(self, x: torch.Tensor) -> torch.Tensor:
        dev = x.device
        B, _, H, W = x.shape
        self._sync_flat()
        key = (B, H, W)
        wp, wb = self._train_workspace(B, H, W, dev)
        keep = self.keep_scale_override
        if keep is None and self.drop_path_rate > 0:
            kp = getattr(self, "_kp_dev", None)                                           # [2 * blocks, 1], cached per device
            if kp is None or kp.device != dev:                                            # (no host copy per step: graph-capturable)
                kp = self._kp_dev = self.drop_path_keep_probs().to(dev).repeat_interleave(2)[:, None].contiguous()
            keep = torch.floor(kp + torch.rand(kp.shape[0], B, device=dev)) / kp         # drct.py:107-119
        self._keep = None if keep is None else keep.to(device=dev, dtype=torch.float32).contiguous()
        s = self.upscale
        y = torch.empty(B, self.cfg.in_chans, H * s, W * s, dtype=torch.float32, device=dev)
        L.check(L.lib().srad_drct_forward_train(self._handle, L.dptr(x), B, H, W, L.dptr(y), L.dptr(self._keep), wp, wb,
                                                L.current_stream_ptr()), "drct_forward_train")
        self._train_shape = key
        return y

    def _backward(self, dy: torch.Tensor, need_dx: bool) -> Optional[torch.Tensor]:
        B, H, W = self._train_shape
        if self._grad_views[0][0].grad is None:          # optimizer.zero_grad(set_to_none=True) dropped the views
            self.zero_grad()
        wp, wb = self._aligned(self._train_ws[(B, H, W)])
        dx = torch.empty(B, self.cfg.in_chans, H, W, dtype=torch.float32, device=dy.device) if need_dx else None
        failed = []

        def hook(user, b):                  # ctypes swallows exceptions raised inside a callback: keep the first one
            try:
                if not failed:
                    self.on_bucket(b)
            except BaseException as e:      # noqa: BLE001 - re-raised below, after the C call has returned
                failed.append(e)
        cb = L.BUCKET_FN(hook) if self.on_bucket is not None else L.BUCKET_FN(0)
        L.check(L.lib().srad_drct_backward(self._handle, L.dptr(dy), B, H, W, L.dptr(self._keep), L.dptr(dx),
                                           L.dptr(self.flat_grads), wp, wb, L.current_stream_ptr(), cb, None), "drct_backward")
        if failed:
            raise RuntimeError(f"gradient bucket hook failed: {failed[0]!r} - gradients of this step are not reduced") from failed[0]
        return dx

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        if self.training and torch.is_grad_enabled():
            if not x.is_cuda:
                raise RuntimeError("srad_amd runs on the GPU only (HIP engine); got a CPU tensor - there is no CPU fallback")
            if not self._can_train():
                raise NotImplementedError(f"DRCT: the HIP backward pass is built for window sizes up to 16 (got {self.window_size}); "
                                          "run larger windows under torch.no_grad() / .eval()")
            if x.dim() != 4 or x.shape[1] != self.cfg.in_chans:
                raise ValueError(f"expected a [B, {self.cfg.in_chans}, H, W] tensor")
            if x.shape[2] % self.window_size or x.shape[3] % self.window_size:
                raise ValueError(f"input {x.shape[2]}x{x.shape[3]} must be a multiple of the window size {self.window_size}")
            if getattr(self, "flat_params", None) is None:
                self.enable_training()
            return _DrctTrainFn.apply(x, self._anchor, self)
        x = self._check_input(x, self.cfg.in_chans)
        dev = x.device
        B, _, H, W = x.shape
        if H % self.window_size or W % self.window_size:
            raise ValueError(f"input {H}x{W} must be a multiple of the window size {self.window_size}")
        self._sync_weights(dev)
        ws = self._workspace(B, H, W, dev)
        s = self.upscale
        if self.use_graph:
            key = (B, H, W)
            if key not in self._static_io:
                self._static_io = {key: (torch.empty_like(x), torch.empty(B, self.cfg.in_chans, H * s, W * s,
                                                                          dtype=torch.float32, device=dev))}
            xin, yout = self._static_io[key]
        else:
            xin = x
            yout = torch.empty(B, self.cfg.in_chans, H * s, W * s, dtype=torch.float32, device=dev)
        wp, wb = self._aligned(ws)

        def launch():
            if self.use_graph:
                xin.copy_(x)
            L.check(L.lib().srad_drct_forward(self._handle, L.dptr(xin), B, H, W, L.dptr(yout), wp, wb,
                                              L.current_stream_ptr()), "drct_forward")
            return yout.clone() if self.use_graph else yout
        return self._run(launch)


# ------------------------------------------------------------------ DRN
class DRN(_EngineModule):
    """Drop-in for reference ``src.drn.DRN(opt)``; ``forward`` returns the list of phase+1 images,
    coarse -> fine (src/drn.py:258-270)."""

    _prefix = "srad_drn"

    def __init__(self, opt, precision: Optional[str] = None, use_graph: Optional[bool] = None):
        super().__init__()
        scales = opt.scale if isinstance(opt.scale, (list, tuple)) else [opt.scale]
        cfg = S.DRNConfig(n_colors=opt.n_colors, scale=max(scales), n_blocks=opt.n_blocks, n_feats=opt.n_feats,
                          negval=float(opt.negval), rgb_range=float(opt.rgb_range))
        self.cfg = cfg
        self.scale = list(scales)
        self.phase = cfg.phase
        prec = precision or getattr(opt, "precision", "fp32")
        graph = getattr(opt, "use_graph", True) if use_graph is None else use_graph
        self._init_engine(S.drn_spec(cfg), cfg, prec, graph)

    def _make_handle(self):
        c = self.cfg
        cc = L.DrnConfig(c.n_colors, c.scale, c.n_blocks, c.n_feats, c.negval, c.rgb_range,
                         L.PRECISIONS[self.precision], int(self.use_graph))
        h = C.c_void_p()
        L.check(L.lib().srad_drn_create(C.byref(cc), C.byref(h)), "drn_create")
        return h

    # -- training (C ABI srad_drn_forward_train / srad_drn_backward) -----------------------------
    def _can_train(self) -> bool:
        return True                             # incl. the x8 preset (n_feats 10: level 0 stored zero-padded to 12 channels)

    def enable_training(self) -> "DRN":
        super().enable_training()
        self.on_bucket = None               # data-parallel hook: callable(bucket index), see GradReducer
        a, b = C.c_int64(), C.c_int64()
        self.grad_buckets = []              # (offset, floats) of the flat gradient buffer, in the order the backward completes them
        for i in range(L.lib().srad_drn_num_buckets(self._handle)):
            L.check(L.lib().srad_drn_bucket_range(self._handle, i, C.byref(a), C.byref(b)), "drn_bucket_range")
            self.grad_buckets.append((a.value, b.value))
        return self

    def _forward_train(self, x: torch.Tensor) -> List[torch.Tensor]:
        dev = x.device
        B, Cc, H, W = x.shape
        self._sync_flat()
        wp, wb = self._train_workspace(B, H, W, dev)
        outs = [torch.empty(B, Cc, H * 2 ** i, W * 2 ** i, dtype=torch.float32, device=dev) for i in range(self.phase + 1)]
        ptrs = (C.c_void_p * len(outs))(*[o.data_ptr() for o in outs])
        L.check(L.lib().srad_drn_forward_train(self._handle, L.dptr(x), B, H, W, ptrs, len(outs), wp, wb,
                                               L.current_stream_ptr()), "drn_forward_train")
        self._train_shape = (B, H, W)
        return outs

    def _backward(self, dys) -> None:
        """dys: one gradient (or None) per output of the last training forward."""
        B, H, W = self._train_shape
        if self._grad_views[0][0].grad is None:
            self.zero_grad()
        wp, wb = self._train_workspace(B, H, W, self.flat_grads.device)
        keep = [None if g is None else g.to(torch.float32).contiguous() for g in dys]
        ptrs = (C.c_void_p * len(keep))(*[0 if g is None else g.data_ptr() for g in keep])
        failed = []

        def hook(user, b):                  # ctypes swallows exceptions raised inside a callback: keep the first one
            try:
                if not failed:
                    self.on_bucket(b)
            except BaseException as e:      # noqa: BLE001 - re-raised below, after the C call has returned
                failed.append(e)
        cb = L.BUCKET_FN(hook) if self.on_bucket is not None else L.BUCKET_FN(0)
        L.check(L.lib().srad_drn_backward(self._handle, ptrs, len(keep), B, H, W, L.dptr(self.flat_grads), wp, wb,
                                          L.current_stream_ptr(), cb, None), "drn_backward")
        if failed:
            raise RuntimeError(f"gradient bucket hook failed: {failed[0]!r} - gradients of this step are not reduced") from failed[0]

    def forward(self, x: torch.Tensor) -> List[torch.Tensor]:
        if self.training and torch.is_grad_enabled() and self._can_train():
            if not x.is_cuda:
                raise RuntimeError("srad_amd runs on the GPU only (HIP engine); got a CPU tensor - there is no CPU fallback")
            if x.dim() != 4 or x.shape[1] != self.cfg.n_colors:
                raise ValueError(f"expected a [B, {self.cfg.n_colors}, H, W] tensor")
            if getattr(self, "flat_params", None) is None:
                self.enable_training()
            return list(_DrnTrainFn.apply(x, self._anchor, self))
        x = self._check_input(x, self.cfg.n_colors)
        dev = x.device
        B, Cc, H, W = x.shape
        self._sync_weights(dev)
        ws = self._workspace(B, H, W, dev)
        key = (B, H, W)
        if self.use_graph:
            if key not in self._static_io:
                outs = [torch.empty(B, Cc, H * 2 ** i, W * 2 ** i, dtype=torch.float32, device=dev)
                        for i in range(self.phase + 1)]
                self._static_io = {key: (torch.empty_like(x), outs)}
            xin, outs = self._static_io[key]
        else:
            xin = x
            outs = [torch.empty(B, Cc, H * 2 ** i, W * 2 ** i, dtype=torch.float32, device=dev)
                    for i in range(self.phase + 1)]
        ptrs = (C.c_void_p * len(outs))(*[o.data_ptr() for o in outs])
        wp, wb = self._aligned(ws)

        def launch():
            if self.use_graph:
                xin.copy_(x)
            L.check(L.lib().srad_drn_forward(self._handle, L.dptr(xin), B, H, W, ptrs, len(outs), wp, wb,
                                             L.current_stream_ptr()), "drn_forward")
            return [o.clone() for o in outs] if self.use_graph else outs
        return self._run(launch)


class DownBlock(nn.Module):
    """Dual regression model, drop-in for reference ``src.model.DownBlock(opt, 2)``
    (src/model.py:8-44): conv3x3 s2 (no bias) + LeakyReLU(negval) + conv3x3 (no bias)."""

    def __init__(self, opt, scale: int = 2, precision: Optional[str] = None):
        super().__init__()
        if scale != 2:
            raise ValueError("the reference only builds dual models with scale 2")
        self.n_colors, self.n_feats, self.negval = opt.n_colors, opt.n_feats, float(opt.negval)
        self.precision = precision or getattr(opt, "precision", "fp32")
        w0 = torch.empty(self.n_feats, self.n_colors, 3, 3)
        w1 = torch.empty(self.n_colors, self.n_feats, 3, 3)
        nn.init.kaiming_uniform_(w0, a=math.sqrt(5))
        nn.init.kaiming_uniform_(w1, a=math.sqrt(5))
        _attach(self, "dual_module.0.0.weight", w0, True)
        _attach(self, "dual_module.1.weight", w1, True)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        if not x.is_cuda:
            raise RuntimeError("srad_amd runs on the GPU only (HIP engine)")
        w0 = self.get_parameter("dual_module.0.0.weight")
        w1 = self.get_parameter("dual_module.1.weight")
        if torch.is_grad_enabled() and (x.requires_grad or w0.requires_grad or w1.requires_grad):
            if x.shape[2] % 2 or x.shape[3] % 2:
                raise NotImplementedError("DownBlock backward needs even image sizes")
            return _DualFn.apply(x, w0, w1, self)
        return self._run_forward(x.detach().to(torch.float32).contiguous(), w0.detach().contiguous(), w1.detach().contiguous())

    def _run_forward(self, x: torch.Tensor, w0: torch.Tensor, w1: torch.Tensor) -> torch.Tensor:
        B, Cc, H, W = x.shape
        y = torch.empty(B, Cc, (H + 1) // 2, (W + 1) // 2, dtype=torch.float32, device=x.device)
        nbytes = C.c_size_t()
        L.check(L.lib().srad_dual_workspace_bytes(B, Cc, H, W, self.n_feats, C.byref(nbytes)), "dual_workspace_bytes")
        ws = torch.empty(nbytes.value + 256, dtype=torch.uint8, device=x.device)
        wp, wb = _EngineModule._aligned(ws)
        L.check(L.lib().srad_dual_forward(L.dptr(w0), L.dptr(w1), Cc, self.n_feats, self.negval, L.dptr(x), B, H, W,
                                          L.dptr(y), wp, wb, L.PRECISIONS[self.precision], L.current_stream_ptr()),
                "dual_forward")
        return y
