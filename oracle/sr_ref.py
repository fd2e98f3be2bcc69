"""ORACLE (test infrastructure only) - CPU fp32 restatement of the reference's SR forward
passes, written as plain functions over a state-dict.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
this file.  The product path (``srad_amd``) never does; it fails loudly when the HIP library
is missing.

Pinned against the reference itself: ``tests/golden/make_golden.py`` imports the reference
modules from /root/reference, loads the same synthetic state-dict, and stores input/output
pairs under ``tests/golden/``; ``tests/test_oracle_golden.py`` checks this file against
them.  The reference ships no golden vectors of its own.

Each function cites the reference lines it follows (paths relative to /root/reference).
"""
from __future__ import annotations

import math
from typing import Dict, List, Sequence

import numpy as np
import torch
import torch.nn.functional as F


def _t(sd, key) -> torch.Tensor:
    v = sd[key]
    return v if isinstance(v, torch.Tensor) else torch.from_numpy(np.asarray(v))


# ------------------------------------------------------------------ window helpers
def window_partition(x: torch.Tensor, ws: int) -> torch.Tensor:
    """src/drct.py:193-204"""
    B, H, W, C = x.shape
    x = x.view(B, H // ws, ws, W // ws, ws, C)
    return x.permute(0, 1, 3, 2, 4, 5).contiguous().view(-1, ws, ws, C)


def window_reverse(win: torch.Tensor, ws: int, H: int, W: int) -> torch.Tensor:
    """src/drct.py:207-220"""
    B = int(win.shape[0] / (H * W / ws / ws))
    x = win.view(B, H // ws, W // ws, ws, ws, -1)
    return x.permute(0, 1, 3, 2, 4, 5).contiguous().view(B, H, W, -1)


def calculate_mask(H: int, W: int, ws: int, shift: int) -> torch.Tensor:
    """src/drct.py:449-470 (values 0 / -100.0, not -inf)"""
    img = torch.zeros((1, H, W, 1))
    cnt = 0
    for hs in (slice(0, -ws), slice(-ws, -shift), slice(-shift, None)):
        for wsl in (slice(0, -ws), slice(-ws, -shift), slice(-shift, None)):
            img[:, hs, wsl, :] = cnt
            cnt += 1
    mw = window_partition(img, ws).view(-1, ws * ws)
    am = mw.unsqueeze(1) - mw.unsqueeze(2)
    return am.masked_fill(am != 0, float(-100.0)).masked_fill(am == 0, float(0.0))


def rel_pos_index(ws: int) -> torch.Tensor:
    """src/drct.py:250-260"""
    ch, cw = torch.arange(ws), torch.arange(ws)
    coords = torch.stack(torch.meshgrid(ch, cw, indexing="ij"))
    flat = torch.flatten(coords, 1)
    rel = (flat[:, :, None] - flat[:, None, :]).permute(1, 2, 0).contiguous()
    rel[:, :, 0] += ws - 1
    rel[:, :, 1] += ws - 1
    rel[:, :, 0] *= 2 * ws - 1
    return rel.sum(-1)


# ------------------------------------------------------------------ DRCT
def attention_core(sd, p: str, x: torch.Tensor, ws: int, heads: int, mask, rnd=None, q_fold: float = 1.0,
                   online_chunk: int = 0) -> torch.Tensor:
    """src/drct.py:271-299 (everything of WindowAttention.forward before ``proj``).  x: [B_, N, C] -> [B_, N, C].
    ``rnd`` (tests of the bf16 kernels only): the rounding the engine's bf16 mode applies to MFMA operands - q (after the
    scale), k, v and the un-normalised probabilities exp(s - max); the row sum stays fp32 and un-rounded, as in the kernels.
    ``q_fold`` (with ``rnd``): a constant the kernel folds into q BEFORE rounding it (the 64 x 64-window kernel works in the
    log2 domain: q carries log2 e); the rounded value is divided back, so the arithmetic is unchanged up to that rounding.
    ``online_chunk`` (with ``rnd``): the rounding points of a streaming-softmax kernel that walks the keys in chunks of this
    many - each chunk's probabilities are exp(s - RUNNING max) when they are rounded, earlier partial sums are rescaled in fp32,
    and the denominator is summed from the ROUNDED probabilities (the 64 x 64-window kernel gets it out of P.V through a column
    of ones in V).  Mathematically the same softmax; only where the bf16 roundings fall differs."""
    B_, N, C = x.shape
    qkv = F.linear(x, _t(sd, p + "qkv.weight"), _t(sd, p + "qkv.bias"))
    qkv = qkv.reshape(B_, N, 3, heads, C // heads).permute(2, 0, 3, 1, 4)
    q, k, v = qkv[0], qkv[1], qkv[2]
    q = q * ((C // heads) ** -0.5)
    out = attention_from_qkv(q, k, v, _t(sd, p + "relative_position_bias_table"), ws, mask, rnd, q_fold, online_chunk)
    return out.transpose(1, 2).reshape(B_, N, C)


def attention_from_qkv(q, k, v, table, ws: int, mask, rnd=None, q_fold: float = 1.0, online_chunk: int = 0) -> torch.Tensor:
    """src/drct.py:282-299: softmax(q k^T + relative position bias + mask) v for q (ALREADY times head_dim^-0.5), k, v of shape
    [B_, heads, N, head_dim] -> [B_, heads, N, head_dim].  The hooks are ``attention_core``'s."""
    B_, heads, N, _ = q.shape
    if rnd is not None:
        q, k, v = rnd(q * q_fold) / q_fold, rnd(k), rnd(v)
    attn = q @ k.transpose(-2, -1)
    idx = rel_pos_index(ws)
    bias = table[idx.view(-1)].view(N, N, -1).permute(2, 0, 1).contiguous()
    attn = attn + bias.unsqueeze(0)
    if mask is not None:
        nW = mask.shape[0]
        attn = attn.view(B_ // nW, nW, heads, N, N) + mask.unsqueeze(1).unsqueeze(0)
        attn = attn.view(-1, heads, N, N)
    if rnd is None:
        attn = torch.softmax(attn, dim=-1)
        out = attn @ v
    elif online_chunk > 0:
        m = torch.full(attn.shape[:-1] + (1,), -1e30)
        num = torch.zeros(attn.shape[:-1] + (v.shape[-1],))
        den = torch.zeros_like(m)
        for c in range(0, N, online_chunk):
            sc = attn[..., c:c + online_chunk]
            mnew = torch.maximum(m, sc.amax(-1, keepdim=True))
            alpha = torch.exp(m - mnew)
            pr = rnd(torch.exp(sc - mnew))
            num = num * alpha + pr @ v[..., c:c + online_chunk, :]
            den = den * alpha + pr.sum(-1, keepdim=True)
            m = mnew
        out = num / den
    else:
        e = torch.exp(attn - attn.amax(-1, keepdim=True))
        out = (rnd(e) @ v) / e.sum(-1, keepdim=True)
    return out


def window_attention(sd, p: str, x: torch.Tensor, ws: int, heads: int, mask, rnd=None, taps=None) -> torch.Tensor:
    """src/drct.py:271-302.  x: [B_, N, C]"""
    x = attention_core(sd, p, x, ws, heads, mask, rnd)
    if taps is not None:
        taps["attn_windows"] = x
    if rnd is not None:
        x = rnd(x)
    return F.linear(x, _t(sd, p + "proj.weight"), _t(sd, p + "proj.bias"))


def swin_block(sd, p: str, x: torch.Tensor, H: int, W: int, ws: int, heads: int,
               shift: int, keep=None, rnd=None, taps=None) -> torch.Tensor:
    """src/drct.py:472-512.  x: [B, H*W, C].  ``keep`` (optional [B] tensor of 0/1 divided
    by keep_prob, or a pair of them: the module's drop_path is CALLED twice per block, drct.py:509-510,
    so the attention and MLP branches draw independent masks) restates DropPath in training mode
    (drct.py:107-119); None = eval.  ``rnd``: operand rounding of the engine's bf16 mode (see ``attention_core``), also
    applied to LayerNorm1/2 outputs, the attention output and GELU(fc1) - the A operands of the following GEMMs.
    ``taps`` receives 'attn' (attention output before proj, token order) and 'x1' (after the first residual)."""
    B, L, C = x.shape
    keep_a, keep_m = keep if isinstance(keep, (tuple, list)) else (keep, keep)
    r = (lambda t: t) if rnd is None else rnd
    shortcut = x
    x = r(F.layer_norm(x, (C,), _t(sd, p + "norm1.weight"), _t(sd, p + "norm1.bias"), 1e-5))
    x = x.view(B, H, W, C)
    if shift > 0:
        x = torch.roll(x, shifts=(-shift, -shift), dims=(1, 2))
    xw = window_partition(x, ws).view(-1, ws * ws, C)
    mask = calculate_mask(H, W, ws, shift) if shift > 0 else None
    wt = {} if taps is not None else None
    aw = window_attention(sd, p + "attn.", xw, ws, heads, mask, rnd, wt)
    if taps is not None:
        a = window_reverse(wt["attn_windows"].view(-1, ws, ws, C), ws, H, W)
        if shift > 0:
            a = torch.roll(a, shifts=(shift, shift), dims=(1, 2))
        taps["attn"] = a.reshape(B, H * W, C)
    x = window_reverse(aw.view(-1, ws, ws, C), ws, H, W)
    if shift > 0:
        x = torch.roll(x, shifts=(shift, shift), dims=(1, 2))
    x = x.view(B, H * W, C)
    if keep_a is not None:
        x = x * keep_a.view(B, 1, 1)
    x = shortcut + x
    if taps is not None:
        taps["x1"] = x
    y = r(F.layer_norm(x, (C,), _t(sd, p + "norm2.weight"), _t(sd, p + "norm2.bias"), 1e-5))
    y = F.linear(y, _t(sd, p + "mlp.fc1.weight"), _t(sd, p + "mlp.fc1.bias"))
    y = r(F.gelu(y))                                # exact erf GELU (drct.py:175,184-190)
    y = F.linear(y, _t(sd, p + "mlp.fc2.weight"), _t(sd, p + "mlp.fc2.bias"))
    if keep_m is not None:
        y = y * keep_m.view(B, 1, 1)
    return x + y


def rdg(sd, p: str, x: torch.Tensor, H: int, W: int, cfg, keeps=None) -> torch.Tensor:
    """src/drct.py:388-396.  x: [B, HW, E]"""
    B = x.shape[0]
    feats = [x]
    table = cfg.block_table()
    out = None
    for k, (d, heads, hidden, shift) in enumerate(table, start=1):
        inp = torch.cat(feats, -1) if len(feats) > 1 else feats[0]
        keep = None if keeps is None else keeps[k - 1]
        y = swin_block(sd, f"{p}swin{k}.", inp, H, W, cfg.window_size, heads, shift, keep)
        y = y.transpose(1, 2).reshape(B, d, H, W)                      # pue
        y = F.conv2d(y, _t(sd, f"{p}adjust{k}.weight"), _t(sd, f"{p}adjust{k}.bias"))
        if k < 5:
            y = F.leaky_relu(y, 0.2)
        y = y.flatten(2).transpose(1, 2)                               # pe
        if k < 5:
            feats.append(y)
        else:
            out = y
    return out * 0.2 + x


def drct_forward(sd, x: torch.Tensor, cfg, keeps=None, taps: Dict[str, torch.Tensor] = None) -> torch.Tensor:
    """src/drct.py:886-898 + forward_features 870-884.  x: NCHW fp32 in [0, rgb_range].
    ``taps`` (optional dict) receives intermediate tensors for stage-level parity tests."""
    x = x.float()
    C = x.shape[1]
    if C == 3:
        mean = torch.tensor((0.4488, 0.4371, 0.4040)).view(1, 3, 1, 1)
    else:
        mean = torch.zeros(1, 1, 1, 1)
    x = (x - mean) * cfg.img_range
    x = F.conv2d(x, _t(sd, "conv_first.weight"), _t(sd, "conv_first.bias"), padding=1)
    shallow = x
    B, E, H, W = x.shape
    t = x.flatten(2).transpose(1, 2)
    t = F.layer_norm(t, (E,), _t(sd, "patch_embed.norm.weight"), _t(sd, "patch_embed.norm.bias"), 1e-5)
    if taps is not None:
        taps["embed"] = t.clone()
    for i in range(cfg.n_rdg):
        t = rdg(sd, f"layers.{i}.", t, H, W, cfg, None if keeps is None else keeps[i])
        if taps is not None and i == 0:
            taps["rdg0"] = t.clone()
    t = F.layer_norm(t, (E,), _t(sd, "norm.weight"), _t(sd, "norm.bias"), 1e-5)
    body = t.transpose(1, 2).reshape(B, E, H, W)
    x = F.conv2d(body, _t(sd, "conv_after_body.weight"), _t(sd, "conv_after_body.bias"), padding=1) + shallow
    x = F.leaky_relu(F.conv2d(x, _t(sd, "conv_before_upsample.0.weight"),
                              _t(sd, "conv_before_upsample.0.bias"), padding=1), 0.01)
    for j in range(int(round(math.log2(cfg.upscale)))):
        x = F.conv2d(x, _t(sd, f"upsample.{2 * j}.weight"), _t(sd, f"upsample.{2 * j}.bias"), padding=1)
        x = F.pixel_shuffle(x, 2)
    x = F.conv2d(x, _t(sd, "conv_last.weight"), _t(sd, "conv_last.bias"), padding=1)
    return x / cfg.img_range + mean


# ------------------------------------------------------------------ DRN
def down_block(sd, p: str, x: torch.Tensor, n_down: int, negval: float) -> torch.Tensor:
    """src/drn.py:83-119 == src/model.py:8-44"""
    for j in range(n_down):
        x = F.leaky_relu(F.conv2d(x, _t(sd, f"{p}dual_module.{j}.0.weight"), None, stride=2, padding=1), negval)
    return F.conv2d(x, _t(sd, f"{p}dual_module.{n_down}.weight"), None, padding=1)


def rcab(sd, p: str, x: torch.Tensor) -> torch.Tensor:
    """src/drn.py:143-158 with CALayer 123-139"""
    r = F.relu(F.conv2d(x, _t(sd, p + "body.0.weight"), _t(sd, p + "body.0.bias"), padding=1))
    r = F.conv2d(r, _t(sd, p + "body.2.weight"), _t(sd, p + "body.2.bias"), padding=1)
    y = r.mean(dim=(2, 3), keepdim=True)
    y = F.relu(F.conv2d(y, _t(sd, p + "body.3.conv_du.0.weight"), _t(sd, p + "body.3.conv_du.0.bias")))
    y = torch.sigmoid(F.conv2d(y, _t(sd, p + "body.3.conv_du.2.weight"), _t(sd, p + "body.3.conv_du.2.bias")))
    return r * y + x


def drn_forward(sd, x: torch.Tensor, cfg) -> List[torch.Tensor]:
    """src/drn.py:241-270.  Returns phase+1 images, coarse -> fine."""
    x = x.float()
    P, nb = cfg.phase, cfg.n_blocks
    x = F.interpolate(x, scale_factor=cfg.scale, mode="bicubic", align_corners=False)
    x = F.conv2d(x, _t(sd, "sub_mean.weight"), _t(sd, "sub_mean.bias"))
    x = F.conv2d(x, _t(sd, "head.weight"), _t(sd, "head.bias"), padding=1)
    copies = []
    for p in range(P):
        copies.append(x)
        x = down_block(sd, f"down.{p}.", x, 1, cfg.negval)
    add = lambda t: F.conv2d(t, _t(sd, "add_mean.weight"), _t(sd, "add_mean.bias"))
    results = [add(F.conv2d(x, _t(sd, "tail.0.weight"), _t(sd, "tail.0.bias"), padding=1))]
    for idx in range(P):
        for b in range(nb):
            x = rcab(sd, f"up_blocks.{idx}.{b}.", x)
        x = F.conv2d(x, _t(sd, f"up_blocks.{idx}.{nb}.0.weight"), _t(sd, f"up_blocks.{idx}.{nb}.0.bias"), padding=1)
        x = F.pixel_shuffle(x, 2)
        x = F.conv2d(x, _t(sd, f"up_blocks.{idx}.{nb + 1}.weight"), _t(sd, f"up_blocks.{idx}.{nb + 1}.bias"))
        x = torch.cat((x, copies[P - idx - 1]), 1)
        results.append(add(F.conv2d(x, _t(sd, f"tail.{idx + 1}.weight"), _t(sd, f"tail.{idx + 1}.bias"), padding=1)))
    return results


def dual_forward(sd, x: torch.Tensor, cfg) -> torch.Tensor:
    """Dual regression model DownBlock(opt, 2) (src/model.py:78-82)."""
    return down_block(sd, "", x, 1, cfg.negval)


# ------------------------------------------------------------------ losses (trainer.py:168-188)
def l1(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    """nn.L1Loss(reduction='mean') (src/loss.py:84)"""
    return (a - b).abs().mean()


def drn_total_loss(sr: Sequence[torch.Tensor], lr: Sequence[torch.Tensor], hr: torch.Tensor,
                   sr2lr: Sequence[torch.Tensor], dual_weight: float = 0.1) -> torch.Tensor:
    """src/trainer.py:168-185.  lr = [LR_x(max), ..., LR_x2] coarse->fine as the loader yields."""
    loss_primary = l1(sr[-1], hr)
    for i in range(1, len(sr)):
        loss_primary = loss_primary + l1(sr[i - 1 - len(sr)], lr[i - len(sr)])
    loss_dual = l1(sr2lr[0], lr[0])
    for i in range(1, len(sr2lr)):
        loss_dual = loss_dual + l1(sr2lr[i], lr[i])
    return loss_primary + dual_weight * loss_dual
