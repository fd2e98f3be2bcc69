"""Architecture + state-dict specifications for DRCT-L and DRN-L, and deterministic
synthetic weights.

The key names, shapes and ORDER reproduce what ``state_dict()`` of the reference modules
returns, so reference checkpoints round-trip:
  * DRCT  - reference src/drct.py:716-898 (``DRCT``), RDG 322-396, SwinTransformerBlock
            398-530, WindowAttention 223-318, Mlp 173-190, Upsample 694-713
  * DRN   - reference src/drn.py:160-270 (``DRN``), RCAB 143-158, CALayer 123-139,
            DownBlock 83-119, Upsampler 55-81, MeanShift 44-52
  * dual  - reference src/model.py:8-44 (``DownBlock(opt, 2)``)

Nothing here touches a GPU; it is pure host bookkeeping shared by the product path, the
oracle and the fixture generator.
"""
from __future__ import annotations

import functools
import math
from collections import OrderedDict
from dataclasses import dataclass, field
from typing import Dict, List, Tuple

import numpy as np


# --------------------------------------------------------------------------------------
# configuration objects (plain data; mirror the fields the reference `opt` carries)
# --------------------------------------------------------------------------------------
@dataclass
class DRCTConfig:
    """Hyper-parameters of DRCT-L as the reference builds it (src/main.py:83-142,
    setup_opt_drct 243-294: ``window_size = img_size // 4``)."""
    in_chans: int = 1
    img_size: int = 32           # LR side the model is built for (mask buffers)
    window_size: int = 8
    upscale: int = 4
    embed_dim: int = 180
    n_rdg: int = 12              # len(depths)
    num_heads: int = 6
    mlp_ratio: float = 2.0
    gc: int = 32
    num_feat: int = 64
    img_range: float = 1.0
    drop_path_rate: float = 0.1
    depth_per_rdg: int = 6       # only used for the stochastic-depth decay rule

    @property
    def shift(self) -> int:
        return self.window_size // 2

    def block_table(self) -> List[Tuple[int, int, int, int]]:
        """(dim, heads, hidden, shift) for swin1..swin5 (reference drct.py:324-373)."""
        out = []
        for k in range(5):
            d = self.embed_dim + k * self.gc
            h = self.num_heads if k == 0 else self.num_heads - (d % self.num_heads)
            ratio = self.mlp_ratio if k < 3 else 1
            out.append((d, h, int(d * ratio), self.shift if k in (1, 3) else 0))
        return out

    def drop_path_probs(self) -> List[float]:
        """Per-RDG drop-path rate: dpr[6*i] of linspace(0, rate, sum(depths))
        (reference drct.py:819,829 and RDG's ``drop_path[0]``, 332)."""
        total = self.n_rdg * self.depth_per_rdg
        if total <= 1:
            return [0.0] * self.n_rdg
        return [float(np.float32(self.drop_path_rate) * np.float32(i * self.depth_per_rdg)
                      / np.float32(total - 1)) for i in range(self.n_rdg)]


@dataclass
class DRNConfig:
    """DRN-L hyper-parameters (reference src/main.py:35-81, setup_opt_drn 144-205)."""
    n_colors: int = 1
    scale: int = 4               # max scale; phases = log2(scale)
    n_blocks: int = 40
    n_feats: int = 20
    negval: float = 0.2
    rgb_range: float = 255.0

    @staticmethod
    def for_scale(scale: int, n_colors: int, rgb_range: float = 255.0) -> "DRNConfig":
        table = {2: (44, 40), 4: (40, 20), 8: (36, 10)}
        nb, nf = table[scale]
        return DRNConfig(n_colors=n_colors, scale=scale, n_blocks=nb, n_feats=nf,
                         rgb_range=rgb_range)

    @property
    def phase(self) -> int:
        return int(round(math.log2(self.scale)))

    @property
    def scales(self) -> List[int]:
        return [2 ** (s + 1) for s in range(self.phase)]


# --------------------------------------------------------------------------------------
# integer / mask buffers (reference drct.py:250-260, 449-470)
# --------------------------------------------------------------------------------------
@functools.lru_cache(maxsize=8)
def _relative_position_index_cached(ws: int) -> np.ndarray:
    a = _relative_position_index(ws)
    a.setflags(write=False)
    return a


def relative_position_index(ws: int) -> np.ndarray:
    """[N, N] int64 index into the (2 ws - 1)^2 bias table (reference drct.py:250-260).  Cached and read-only: at
    window 64 it is 134 MB and every one of the 60 blocks registers the same buffer."""
    return _relative_position_index_cached(int(ws))


def _relative_position_index(ws: int) -> np.ndarray:
    coords = np.stack(np.meshgrid(np.arange(ws), np.arange(ws), indexing="ij"))  # 2,ws,ws
    flat = coords.reshape(2, -1)
    rel = flat[:, :, None] - flat[:, None, :]
    rel = rel.transpose(1, 2, 0).copy()
    rel[:, :, 0] += ws - 1
    rel[:, :, 1] += ws - 1
    rel[:, :, 0] *= 2 * ws - 1
    return rel.sum(-1).astype(np.int64)


@functools.lru_cache(maxsize=4)
def _shifted_window_mask_cached(H: int, W: int, ws: int, shift: int) -> np.ndarray:
    a = _shifted_window_mask(H, W, ws, shift)
    a.setflags(write=False)
    return a


def shifted_window_mask(H: int, W: int, ws: int, shift: int) -> np.ndarray:
    """[nW, N, N] float32 mask with values 0 / -100 (reference drct.py:449-470).  Cached and read-only (1 GB at
    window 64 on a 256 x 256 image)."""
    return _shifted_window_mask_cached(int(H), int(W), int(ws), int(shift))


def _shifted_window_mask(H: int, W: int, ws: int, shift: int) -> np.ndarray:
    img = np.zeros((H, W), dtype=np.float32)
    cnt = 0
    for hs in (slice(0, -ws), slice(-ws, -shift), slice(-shift, None)):
        for wsl in (slice(0, -ws), slice(-ws, -shift), slice(-shift, None)):
            img[hs, wsl] = cnt
            cnt += 1
    mw = img.reshape(H // ws, ws, W // ws, ws).transpose(0, 2, 1, 3).reshape(-1, ws * ws)
    diff = mw[:, None, :] - mw[:, :, None]
    return np.where(diff != 0, np.float32(-100.0), np.float32(0.0)).astype(np.float32)


# --------------------------------------------------------------------------------------
# state-dict specs:  name -> (shape, kind)
# kinds: conv_w, conv_b, lin_w, lin_b, ln_w, ln_b, table, index, mask, meanshift_w,
#        meanshift_b_sub, meanshift_b_add
# --------------------------------------------------------------------------------------
Spec = "OrderedDict[str, Tuple[Tuple[int, ...], str]]"


def drct_spec(cfg: DRCTConfig) -> "OrderedDict[str, Tuple[Tuple[int, ...], str]]":
    s: "OrderedDict[str, Tuple[Tuple[int, ...], str]]" = OrderedDict()
    E, C, ws = cfg.embed_dim, cfg.in_chans, cfg.window_size
    N = ws * ws
    nW = (cfg.img_size // ws) ** 2
    s["conv_first.weight"] = ((E, C, 3, 3), "conv_w")
    s["conv_first.bias"] = ((E,), "conv_b")
    s["patch_embed.norm.weight"] = ((E,), "ln_w")
    s["patch_embed.norm.bias"] = ((E,), "ln_b")
    for i in range(cfg.n_rdg):
        for k, (d, h, m, shift) in enumerate(cfg.block_table(), start=1):
            p = f"layers.{i}.swin{k}."
            if shift > 0:
                s[p + "attn_mask"] = ((nW, N, N), "mask")
            s[p + "norm1.weight"] = ((d,), "ln_w")
            s[p + "norm1.bias"] = ((d,), "ln_b")
            s[p + "attn.relative_position_bias_table"] = (((2 * ws - 1) ** 2, h), "table")
            s[p + "attn.relative_position_index"] = ((N, N), "index")
            s[p + "attn.qkv.weight"] = ((3 * d, d), "lin_w")
            s[p + "attn.qkv.bias"] = ((3 * d,), "lin_b")
            s[p + "attn.proj.weight"] = ((d, d), "lin_w")
            s[p + "attn.proj.bias"] = ((d,), "lin_b")
            s[p + "norm2.weight"] = ((d,), "ln_w")
            s[p + "norm2.bias"] = ((d,), "ln_b")
            s[p + "mlp.fc1.weight"] = ((m, d), "lin_w")
            s[p + "mlp.fc1.bias"] = ((m,), "lin_b")
            s[p + "mlp.fc2.weight"] = ((d, m), "lin_w")
            s[p + "mlp.fc2.bias"] = ((d,), "lin_b")
            out = cfg.gc if k < 5 else E
            s[f"layers.{i}.adjust{k}.weight"] = ((out, d, 1, 1), "conv_w")
            s[f"layers.{i}.adjust{k}.bias"] = ((out,), "conv_b")
    s["norm.weight"] = ((E,), "ln_w")
    s["norm.bias"] = ((E,), "ln_b")
    s["conv_after_body.weight"] = ((E, E, 3, 3), "conv_w")
    s["conv_after_body.bias"] = ((E,), "conv_b")
    F = cfg.num_feat
    s["conv_before_upsample.0.weight"] = ((F, E, 3, 3), "conv_w")
    s["conv_before_upsample.0.bias"] = ((F,), "conv_b")
    for j in range(int(round(math.log2(cfg.upscale)))):
        s[f"upsample.{2 * j}.weight"] = ((4 * F, F, 3, 3), "conv_w")
        s[f"upsample.{2 * j}.bias"] = ((4 * F,), "conv_b")
    s["conv_last.weight"] = ((C, F, 3, 3), "conv_w")
    s["conv_last.bias"] = ((C,), "conv_b")
    return s


def drn_spec(cfg: DRNConfig) -> "OrderedDict[str, Tuple[Tuple[int, ...], str]]":
    s: "OrderedDict[str, Tuple[Tuple[int, ...], str]]" = OrderedDict()
    C, F, P, nb = cfg.n_colors, cfg.n_feats, cfg.phase, cfg.n_blocks
    s["sub_mean.weight"] = ((C, C, 1, 1), "meanshift_w")
    s["sub_mean.bias"] = ((C,), "meanshift_b_sub")
    s["add_mean.weight"] = ((C, C, 1, 1), "meanshift_w")
    s["add_mean.bias"] = ((C,), "meanshift_b_add")
    s["head.weight"] = ((F, C, 3, 3), "conv_w")
    s["head.bias"] = ((F,), "conv_b")
    for p in range(P):
        f = F * 2 ** p
        s[f"down.{p}.dual_module.0.0.weight"] = ((f, f, 3, 3), "conv_w")
        s[f"down.{p}.dual_module.1.weight"] = ((2 * f, f, 3, 3), "conv_w")
    top = F * 2 ** P
    for idx in range(P):
        # RCAB width: n_feats*2^phase for the first stack, then n_feats*2^p for
        # p = phase..2 (reference drn.py:199-210)
        ch = top if idx == 0 else F * 2 ** (P - idx + 1)
        for b in range(nb):
            q = f"up_blocks.{idx}.{b}.body."
            s[q + "0.weight"] = ((ch, ch, 3, 3), "conv_w")
            s[q + "0.bias"] = ((ch,), "conv_b")
            s[q + "2.weight"] = ((ch, ch, 3, 3), "conv_w")
            s[q + "2.bias"] = ((ch,), "conv_b")
            s[q + "3.conv_du.0.weight"] = ((ch // 16, ch, 1, 1), "conv_w")
            s[q + "3.conv_du.0.bias"] = ((ch // 16,), "conv_b")
            s[q + "3.conv_du.2.weight"] = ((ch, ch // 16, 1, 1), "conv_w")
            s[q + "3.conv_du.2.bias"] = ((ch,), "conv_b")
        # Upsampler(2, cin) + 1x1 reducing conv (reference drn.py:213-223)
        cin = top if idx == 0 else 2 * F * 2 ** (P - idx)
        cout = F * 2 ** (P - idx - 1)
        s[f"up_blocks.{idx}.{nb}.0.weight"] = ((4 * cin, cin, 3, 3), "conv_w")
        s[f"up_blocks.{idx}.{nb}.0.bias"] = ((4 * cin,), "conv_b")
        s[f"up_blocks.{idx}.{nb + 1}.weight"] = ((cout, cin, 1, 1), "conv_w")
        s[f"up_blocks.{idx}.{nb + 1}.bias"] = ((cout,), "conv_b")
    s["tail.0.weight"] = ((C, top, 3, 3), "conv_w")
    s["tail.0.bias"] = ((C,), "conv_b")
    for j, p in enumerate(range(P, 0, -1), start=1):
        s[f"tail.{j}.weight"] = ((C, F * 2 ** p, 3, 3), "conv_w")
        s[f"tail.{j}.bias"] = ((C,), "conv_b")
    return s


def dual_spec(cfg: DRNConfig) -> "OrderedDict[str, Tuple[Tuple[int, ...], str]]":
    """One dual regression model = DownBlock(opt, 2) with defaults (reference model.py:78-82)."""
    s: "OrderedDict[str, Tuple[Tuple[int, ...], str]]" = OrderedDict()
    s["dual_module.0.0.weight"] = ((cfg.n_feats, cfg.n_colors, 3, 3), "conv_w")
    s["dual_module.1.weight"] = ((cfg.n_colors, cfg.n_feats, 3, 3), "conv_w")
    return s


DRN_GRAY_MEAN = (0.4440,)
DRN_RGB_MEAN = (0.4488, 0.4371, 0.4040)
DRCT_RGB_MEAN = (0.4488, 0.4371, 0.4040)


# --------------------------------------------------------------------------------------
# deterministic synthetic tensors (counter-based; independent of numpy's Generator streams)
# --------------------------------------------------------------------------------------
_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _fnv1a(name: str, seed: int) -> np.uint64:
    h = 0xCBF29CE484222325 ^ (seed & 0xFFFFFFFFFFFFFFFF)
    for b in name.encode():
        h ^= b
        h = (h * 0x100000001B3) & 0xFFFFFFFFFFFFFFFF
    return np.uint64(h)


def _splitmix64(x: np.ndarray) -> np.ndarray:
    with np.errstate(over="ignore"):
        x = x + np.uint64(0x9E3779B97F4A7C15)
        z = x
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


def det_uniform(name: str, n: int, seed: int = 0) -> np.ndarray:
    """n float64 uniforms in (0,1), a pure function of (name, seed, index)."""
    base = _fnv1a(name, seed)
    with np.errstate(over="ignore"):
        ctr = base + np.arange(n, dtype=np.uint64) * np.uint64(0x2545F4914F6CDD1D)
    bits = _splitmix64(ctr) >> np.uint64(11)            # 53 random bits
    return (bits.astype(np.float64) + 0.5) / float(1 << 53)


def det_normal(name: str, shape, seed: int = 0) -> np.ndarray:
    n = int(np.prod(shape)) if len(shape) else 1
    u1 = det_uniform(name + "#a", n, seed)
    u2 = det_uniform(name + "#b", n, seed)
    z = np.sqrt(-2.0 * np.log(u1)) * np.cos(2.0 * np.pi * u2)
    return z.reshape(shape)


def synth_tensor(name: str, shape, kind: str, seed: int = 0, gain: float = 1.0,
                 cfg=None) -> np.ndarray:
    """Synthetic value for one state-dict entry.  Scales are chosen so activations stay O(1)
    through the whole network and every term (biases, LN affine, bias table, mask) matters
    numerically - unlike the reference's init (Linear std .02, zero biases) under which a
    wrong attention bias would hide below the tolerance."""
    if kind in ("index", "mask"):
        a = (relative_position_index(cfg.window_size) if kind == "index"
             else shifted_window_mask(cfg.img_size, cfg.img_size, cfg.window_size, cfg.shift))
        return a.copy() if a.nbytes < (1 << 24) else a      # big ones (window >= 32) stay shared and read-only
    if kind == "meanshift_w":
        return np.eye(shape[0], dtype=np.float32).reshape(shape)
    if kind in ("meanshift_b_sub", "meanshift_b_add"):
        mean = DRN_GRAY_MEAN if shape[0] == 1 else DRN_RGB_MEAN
        sign = -1.0 if kind.endswith("sub") else 1.0
        return (sign * cfg.rgb_range * np.asarray(mean, dtype=np.float32)).astype(np.float32)
    z = det_normal(name, shape, seed)
    if kind in ("conv_w", "lin_w"):
        fan_in = int(np.prod(shape[1:]))
        v = z * (gain / math.sqrt(fan_in))
    elif kind in ("conv_b", "lin_b"):
        v = z * 0.05
    elif kind == "ln_w":
        v = 1.0 + 0.1 * z
    elif kind == "ln_b":
        v = 0.05 * z
    elif kind == "table":
        v = 0.5 * z
    else:
        raise ValueError(kind)
    return v.astype(np.float32)


def synth_state(spec, seed: int = 0, gain: float = 1.0, cfg=None) -> "OrderedDict[str, np.ndarray]":
    return OrderedDict((k, synth_tensor(k, shp, kind, seed, gain, cfg))
                       for k, (shp, kind) in spec.items())


def synth_image(name: str, shape, seed: int = 1, rgb_range: float = 255.0) -> np.ndarray:
    """LR tile in [0, rgb_range] (value range of the reference loader, data.py:11-17)."""
    n = int(np.prod(shape))
    return (det_uniform(name, n, seed).reshape(shape) * rgb_range).astype(np.float32)


# -------------------------------------------------------------------------------------- synthetic scorer inputs
def synth_pairs(n_good: int, n_bad: int, size: int, channels: int = 1, seed: int = 0
                ) -> Tuple[List[int], List[np.ndarray], List[np.ndarray]]:
    """Grid-textured u8 HR images; SR = HR + noise, with planted blobs in the 'bad' half."""
    rng = np.random.RandomState(seed)
    yy, xx = np.mgrid[0:size, 0:size].astype(np.float32)
    y_true, sr, hr = [], [], []
    for i in range(n_good + n_bad):
        period = 6 + (i % 5)
        tex = 120 + 60 * np.sin(2 * np.pi * xx / period) * np.sin(2 * np.pi * yy / period)
        tex = tex[..., None] + rng.normal(0, 4, (size, size, channels))
        h = np.clip(tex, 0, 255).astype(np.uint8)
        s = h.astype(np.float32) + rng.normal(0, 3, h.shape)
        bad = i >= n_good
        if bad:
            cy, cx = rng.randint(size // 4, 3 * size // 4, 2)
            r = max(2, size // 12)
            blob = ((yy - cy) ** 2 + (xx - cx) ** 2) < r * r
            s[blob] += rng.uniform(15, 50)
        y_true.append(int(bad))
        sr.append(np.clip(s, 0, 255).astype(np.uint8))
        hr.append(h)
    return y_true, sr, hr
