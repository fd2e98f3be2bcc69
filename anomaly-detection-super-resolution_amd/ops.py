"""Single-operator bindings (C ABI ``srad_op_*``).  Activations are NHWC / token-major fp32
tensors on the GPU; weights are given in PyTorch layout and packed by the library."""
from __future__ import annotations

import ctypes as C
from typing import Optional, Tuple

import torch

from . import _lib as L


def _need_cuda(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise RuntimeError("srad_amd ops run on the GPU only (HIP); there is no CPU fallback")


def gemm(x: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor] = None, *, B: int = 1, H: int = 0,
         W: int = 0, stride: int = 1, ln: Optional[tuple] = None, act: int = L.ACT_NONE, slope: float = 0.0,
         alpha: float = 1.0, residual: Optional[torch.Tensor] = None, out: Optional[torch.Tensor] = None,
         out_offset: int = 0, pixel_shuffle: bool = False, precision: str = "fp32") -> torch.Tensor:
    """x: [rows, Cin] (row stride = x.stride(0)); weight: [N, Cin] or [N, Cin, 3, 3].
    For a 3x3 / strided conv pass the image geometry (B, H, W) of the NHWC input rows."""
    _need_cuda(x, weight)
    assert x.dim() == 2 and (x.stride(1) == 1 or x.shape[1] == 1) and x.dtype == torch.float32
    ldx = x.stride(0) if x.shape[0] > 1 else x.shape[1]
    ntaps = 9 if weight.dim() == 4 and weight.shape[-1] == 3 else 1
    N, Cin = weight.shape[0], weight.shape[1]
    assert x.shape[1] >= Cin
    w = weight.detach().reshape(N, Cin, ntaps).float()
    if Cin % 4 or ldx % 4 or x.data_ptr() % 16:
        # the kernels address activations as float4: pad the channels (zeros) like the engines do
        cpad = (Cin + 3) // 4 * 4
        x = torch.nn.functional.pad(x[:, :Cin], (0, cpad - Cin)).contiguous()
        w = torch.nn.functional.pad(w, (0, 0, 0, cpad - Cin))
        if ln is not None:
            raise ValueError("LayerNorm fusion needs a channel count that is a multiple of 4")
        Cin, ldx = cpad, cpad
    w = w.contiguous()
    if ntaps == 1 and stride == 1:
        B, H, W = 1, 1, x.shape[0]
    pad, k = (1, 3) if ntaps == 9 else (0, 1)
    Ho, Wo = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
    M = B * Ho * Wo
    if out is None:
        if pixel_shuffle:
            out = torch.empty(M * 4, N // 4, dtype=torch.float32, device=x.device)
        else:
            out = torch.empty(M, N, dtype=torch.float32, device=x.device)
    prec = L.PRECISIONS[precision]
    nbytes = L.lib().srad_op_gemm_scratch_bytes(prec, N, Cin, ntaps)
    scratch = torch.empty(nbytes + 256, dtype=torch.uint8, device=x.device)
    off = (-scratch.data_ptr()) % 256
    g, b_ = (ln if ln is not None else (None, None))
    L.check(L.lib().srad_op_gemm(prec, L.dptr(x), ldx, B, H, W, Cin, L.dptr(w), N, ntaps, stride,
                                 L.dptr(bias), L.dptr(g), L.dptr(b_), act, slope, alpha, L.dptr(residual),
                                 0 if residual is None else residual.stride(0), L.dptr(out), out.stride(0),
                                 out_offset, 2 if pixel_shuffle else 0, C.c_void_p(scratch.data_ptr() + off),
                                 C.c_size_t(nbytes), L.current_stream_ptr()), "op_gemm")
    return out


def window_attention(qkv: torch.Tensor, table: torch.Tensor, B: int, H: int, W: int, ws: int, shift: int,
                     heads: int, precision: str = "fp32", pad_value: float = 0.0) -> torch.Tensor:
    """qkv: [B*H*W, 3d] raster-ordered tokens in the reference's column order (q | k | v, heads
    contiguous) -> [B*H*W, d].  The kernel reads a head-padded layout (every head slice padded to a
    multiple of 4 floats; the engine's QKV GEMM writes it directly), so this wrapper re-lays it out."""
    _need_cuda(qkv, table)
    assert qkv.shape[0] == B * H * W
    d = qkv.shape[1] // 3
    hd = d // heads
    hdp = (hd + 3) // 4 * 4
    padded = torch.nn.functional.pad(qkv.reshape(-1, 3 * heads, hd).float(), (0, hdp - hd), value=pad_value)
    padded = padded.reshape(-1, 3 * heads * hdp).contiguous()   # pad columns are never read as data (any value, NaN too)
    out = torch.empty(qkv.shape[0], d, dtype=torch.float32, device=qkv.device)
    L.check(L.lib().srad_op_window_attn(L.PRECISIONS[precision], L.dptr(padded), L.dptr(out),
                                        L.dptr(table.contiguous()), B, H, W, ws, shift, d, heads, hdp,
                                        L.current_stream_ptr()), "op_window_attn")
    return out


def layernorm(x: torch.Tensor, g: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    _need_cuda(x, g, b)
    assert x.dim() == 2 and x.stride(1) == 1
    y = torch.empty(x.shape, dtype=torch.float32, device=x.device)
    L.check(L.lib().srad_op_layernorm(L.dptr(x), x.stride(0), L.dptr(y), y.stride(0), x.shape[0], x.shape[1],
                                      L.dptr(g), L.dptr(b), L.current_stream_ptr()), "op_layernorm")
    return y


# ----------------------------------------------------------------------------------- backward operators
_WGRAD_WS = {}


def wgrad_workspace(device) -> C.c_void_p:
    """Zeroed-once split-K workspace of the weight-gradient kernel, one per device."""
    key = str(device)
    if key not in _WGRAD_WS:
        n = L.lib().srad_op_wgrad_workspace_bytes()
        _WGRAD_WS[key] = torch.zeros(n + 256, dtype=torch.uint8, device=device)
    t = _WGRAD_WS[key]
    return C.c_void_p(t.data_ptr() + (-t.data_ptr()) % 256)


def wgrad(dy: torch.Tensor, x: torch.Tensor, N: int, Cin: int, *, ntaps: int = 1, B: int = 1, H: int = 0, W: int = 0,
          stride: int = 1, row_scale: Optional[torch.Tensor] = None, alpha: float = 1.0, bias: bool = True,
          precision: str = "fp32"):
    """Weight / bias gradient of Linear (ntaps 1) or a 3x3 conv: dy [B*Ho*Wo, >=N], x [B*H*W, >=Cin] NHWC rows.
    Returns (dW [N, Cin, ntaps], db [N] or None), freshly zeroed and accumulated into by the kernel."""
    _need_cuda(dy, x)
    assert dy.stride(1) == 1 and x.stride(1) == 1 and N % 4 == 0 and Cin % 4 == 0
    if ntaps == 1 and stride == 1:
        B, H, W = 1, 1, x.shape[0]
    dw = torch.zeros(N, Cin, ntaps, dtype=torch.float32, device=x.device)
    db = torch.zeros(N, dtype=torch.float32, device=x.device) if bias else None
    L.check(L.lib().srad_op_wgrad(L.PRECISIONS[precision], L.dptr(dy), dy.stride(0), L.dptr(x), x.stride(0), B, H, W, N,
                                  Cin, ntaps, stride, L.dptr(row_scale), alpha, L.dptr(dw), L.dptr(db),
                                  wgrad_workspace(x.device), L.current_stream_ptr()), "op_wgrad")
    return dw, db


def conv80_bf16(x_h: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor] = None, *, B: int, H: int, W: int, act: int = 0,
                slope: float = 0.0, residual: Optional[torch.Tensor] = None, rmode: int = 0, out_bf16: bool = True,
                pool: bool = False):
    """The 80 -> 80 channel 3x3 convolution on a bf16 activation array, as DRN's bf16 chains run it (C ABI ``srad_op_conv80_h``):
    ``x_h`` [B*H*W, 80] bf16 NHWC rows, ``weight`` [80, 80, 3, 3]; ``residual`` bf16 or fp32 [B*H*W, 80] (rmode 0: added; 2: the
    output is multiplied by (residual > 0 ? 1 : slope)); output bf16 or fp32.  ``pool=True`` also returns the per-tile column
    sums [B*H*W/128, 80]."""
    _need_cuda(x_h, weight)
    assert x_h.dtype == torch.bfloat16 and x_h.is_contiguous() and x_h.shape == (B * H * W, 80)
    w = weight.detach().reshape(80, 80, 9).float().contiguous()
    y = torch.empty(B * H * W, 80, dtype=torch.bfloat16 if out_bf16 else torch.float32, device=x_h.device)
    part = torch.empty(B * H * W // 128, 80, dtype=torch.float32, device=x_h.device) if pool else None
    prec = L.PRECISIONS["bf16"]
    nbytes = L.lib().srad_op_gemm_scratch_bytes(prec, 80, 80, 9)
    scratch = torch.empty(nbytes + 256, dtype=torch.uint8, device=x_h.device)
    sp = C.c_void_p(scratch.data_ptr() + (-scratch.data_ptr()) % 256)
    r_h = residual if residual is not None and residual.dtype == torch.bfloat16 else None
    r_f = residual if residual is not None and residual.dtype == torch.float32 else None
    L.check(L.lib().srad_op_conv80_h(L.dptr(x_h), L.dptr(w), L.dptr(bias), act, slope, L.dptr(r_h), L.dptr(r_f), rmode, B, H, W,
                                     L.dptr(y) if out_bf16 else None, None if out_bf16 else L.dptr(y), L.dptr(part), sp, nbytes,
                                     L.current_stream_ptr()), "op_conv80_h")
    return (y, part) if pool else y


def wgrad_conv9_bf16(dy_h: torch.Tensor, x: torch.Tensor, *, B: int, H: int, W: int):
    """Weight / bias gradient of that convolution from a bf16 dY and a bf16 or fp32 X (C ABI ``srad_op_wgrad_conv9_h``):
    returns (dW [80, 80, 9], db [80])."""
    _need_cuda(dy_h, x)
    assert dy_h.dtype == torch.bfloat16 and dy_h.is_contiguous() and x.is_contiguous() and dy_h.shape == x.shape == (B * H * W, 80)
    dw = torch.zeros(80, 80, 9, dtype=torch.float32, device=x.device)
    db = torch.zeros(80, dtype=torch.float32, device=x.device)
    L.check(L.lib().srad_op_wgrad_conv9_h(L.dptr(dy_h), L.dptr(x), int(x.dtype == torch.bfloat16), B, H, W, L.dptr(dw), L.dptr(db),
                                          wgrad_workspace(x.device), L.current_stream_ptr()), "op_wgrad_conv9_h")
    return dw, db


def dgrad(dy: torch.Tensor, weight: torch.Tensor, *, B: int = 1, H: int = 0, W: int = 0, r: Optional[torch.Tensor] = None,
          rmode: int = 0, slope: float = 0.0, alpha: float = 1.0, row_scale: Optional[torch.Tensor] = None,
          precision: str = "fp32") -> torch.Tensor:
    """Data gradient dx = dy . W of Linear / 3x3 stride-1 conv (weight [N, Cin] or [N, Cin, 3, 3])."""
    _need_cuda(dy, weight)
    ntaps = 9 if weight.dim() == 4 and weight.shape[-1] == 3 else 1
    N, Cin = weight.shape[0], weight.shape[1]
    w = weight.detach().reshape(N, Cin, ntaps).float().contiguous()
    if ntaps == 1:
        B, H, W = 1, 1, dy.shape[0]
    dx = torch.empty(dy.shape[0], Cin, dtype=torch.float32, device=dy.device)
    prec = L.PRECISIONS[precision]
    nbytes = L.lib().srad_op_gemm_scratch_bytes(prec, Cin, N, ntaps)
    scratch = torch.empty(nbytes + 256, dtype=torch.uint8, device=dy.device)
    off = (-scratch.data_ptr()) % 256
    L.check(L.lib().srad_op_dgrad(prec, L.dptr(dy), dy.stride(0), B, H, W, N, L.dptr(w), Cin, ntaps, L.dptr(r),
                                  0 if r is None else r.stride(0), rmode, slope, alpha, L.dptr(row_scale), L.dptr(dx),
                                  dx.stride(0), C.c_void_p(scratch.data_ptr() + off), C.c_size_t(nbytes),
                                  L.current_stream_ptr()), "op_dgrad")
    return dx


def layernorm_bwd(dxn: torch.Tensor, x: torch.Tensor, gamma: torch.Tensor, dres: Optional[torch.Tensor] = None,
                  out: Optional[torch.Tensor] = None):
    """Returns (dx (+ dres, + out if given), dgamma, dbeta)."""
    _need_cuda(dxn, x, gamma)
    rows, Cc = dxn.shape
    acc = out is not None
    if out is None:
        out = torch.empty(rows, Cc, dtype=torch.float32, device=x.device)
    dg = torch.zeros(Cc, dtype=torch.float32, device=x.device)
    db = torch.zeros(Cc, dtype=torch.float32, device=x.device)
    L.check(L.lib().srad_op_layernorm_bwd(L.dptr(dxn.contiguous()), L.dptr(x), x.stride(0), L.dptr(gamma),
                                          L.dptr(None if dres is None else dres.contiguous()), L.dptr(out), 1 if acc else 0,
                                          L.dptr(dg), L.dptr(db), rows, Cc, wgrad_workspace(x.device),
                                          L.current_stream_ptr()), "op_layernorm_bwd")
    return out, dg, db


def window_attention_bwd(qkv: torch.Tensor, dout: torch.Tensor, table: torch.Tensor, B: int, H: int, W: int, ws: int,
                         shift: int, heads: int, precision: str = "fp32"):
    """Backward of window_attention: qkv [T, 3d] (reference column order), dout [T, d] -> (dqkv [T, 3d], dtable)."""
    _need_cuda(qkv, dout, table)
    d = qkv.shape[1] // 3
    hd = d // heads
    hdp = (hd + 3) // 4 * 4
    padded = torch.nn.functional.pad(qkv.reshape(-1, 3 * heads, hd).float(), (0, hdp - hd)).reshape(-1, 3 * heads * hdp).contiguous()
    dqkv = torch.empty(qkv.shape[0], 3 * d, dtype=torch.float32, device=qkv.device)
    dtable = torch.zeros_like(table, dtype=torch.float32).contiguous()
    L.check(L.lib().srad_op_window_attn_bwd(L.PRECISIONS[precision], L.dptr(padded), L.dptr(dout.contiguous()), L.dptr(dqkv), L.dptr(table.contiguous()),
                                            L.dptr(dtable), B, H, W, ws, shift, d, heads, hdp, wgrad_workspace(qkv.device),
                                            L.current_stream_ptr()),
            "op_window_attn_bwd")
    return dqkv, dtable


def wgrad_linear_deferred(dy: torch.Tensor, x: torch.Tensor, row_scale: Optional[torch.Tensor] = None, rps: int = 0,
                          alpha: float = 1.0, bias: bool = True, precision: str = "bf16"):
    """dW [N, Cin] (and db [N]) of a Linear layer from dy [M, N] and x [M, Cin], issued the way the training step does
    (deferred launch + reduce).  fp32 or bf16 operand tensors (bf16 dy: already scaled, needs a bf16 x)."""
    _need_cuda(dy, x)
    M, N = dy.shape
    Cin = x.shape[1]
    dw = torch.zeros(N, Cin, dtype=torch.float32, device=dy.device)
    db = torch.zeros(N, dtype=torch.float32, device=dy.device) if bias else None
    dy, x = dy.contiguous(), x.contiguous()
    L.check(L.lib().srad_op_wgrad_deferred(L.PRECISIONS[precision], L.dptr(dy), N, int(dy.dtype == torch.bfloat16), L.dptr(x), Cin,
                                           int(x.dtype == torch.bfloat16), M, N, Cin, L.dptr(row_scale), rps, alpha, L.dptr(dw), L.dptr(db),
                                           wgrad_workspace(dy.device), L.current_stream_ptr()), "op_wgrad_deferred")
    return dw, db


def window_attention_bwd_bf16io(qkv: torch.Tensor, dout: torch.Tensor, table: torch.Tensor, B: int, H: int, W: int, ws: int,
                                shift: int, heads: int, pad_fill: float = float("nan")):
    """The training step's form of ``window_attention_bwd`` (head dim <= 128): the operands cross the boundary as bf16 in
    per-head slots of ``hp = ceil8(head dim)`` columns - q already scaled, as the fused forward saves it - and dq | dk | dv
    come back as bf16.  ``pad_fill`` goes into dO's padding columns, which the kernel must ignore."""
    _need_cuda(qkv, dout, table)
    d = qkv.shape[1] // 3
    hd = d // heads
    hp = (hd + 7) // 8 * 8
    T = qkv.shape[0]
    q3 = qkv.reshape(T, 3, heads, hd).float().clone()
    q3[:, 0] *= hd ** -0.5
    qkv_h = torch.zeros(T, 3, heads, hp, dtype=torch.bfloat16, device=qkv.device)
    qkv_h[..., :hd] = q3.to(torch.bfloat16)
    dout_h = torch.full((T, heads, hp), pad_fill, dtype=torch.bfloat16, device=qkv.device)
    dout_h[..., :hd] = dout.reshape(T, heads, hd).to(torch.bfloat16)
    dqkv_h = torch.empty(T, 3 * d, dtype=torch.bfloat16, device=qkv.device)
    dtable = torch.zeros_like(table, dtype=torch.float32).contiguous()
    L.check(L.lib().srad_op_window_attn_bwd_h(L.dptr(qkv_h), L.dptr(dout_h), L.dptr(dqkv_h), L.dptr(table.contiguous()), L.dptr(dtable),
                                              B, H, W, ws, shift, d, heads, hp, wgrad_workspace(qkv.device), L.current_stream_ptr()),
            "op_window_attn_bwd_h")
    return dqkv_h.float(), dtable


def _scratch(nbytes: int, device) -> Tuple[torch.Tensor, C.c_void_p, C.c_size_t]:
    t = torch.empty(nbytes + 256, dtype=torch.uint8, device=device)
    off = (-t.data_ptr()) % 256
    return t, C.c_void_p(t.data_ptr() + off), C.c_size_t(nbytes)


def mlp_bwd(dx2: Optional[torch.Tensor], hpre: torch.Tensor, x1: torch.Tensor, gamma: torch.Tensor, w_fc1: torch.Tensor,
            w_fc2: torch.Tensor, rs2: Optional[torch.Tensor] = None, rps: int = 0, adjust=None, proj=None):
    """Fused backward of a Swin block's MLP branch (bf16 MFMA, C ABI ``srad_op_mlp_bwd``; src/drct.py:510, 184-190).

    ``dx2`` [M, d] is the gradient of ``x2 = x1 + rs2 * fc2(gelu(fc1(LN(x1))))``, or None when ``adjust`` =
    ``(dA [M, KA], y_act [M, KA] or None, slope, alpha, w_adj [KA, d])`` is given: then dx2 = alpha * (dA * lrelu'(y_act)) @ w_adj
    is computed first.  ``proj`` = ``(w_proj [d, d], rs1 or None)`` adds dO = (dx1 @ w_proj) * rs1.
    Returns a dict: dh [M, m], dx1 [M, d], dgamma, dbeta, and dx2 / dA (adjust) / dO (proj)."""
    _need_cuda(hpre, x1, gamma, w_fc1, w_fc2)
    M, d = x1.shape
    m = hpre.shape[1]
    dev = x1.device
    f = lambda t: None if t is None else t.detach().float().contiguous()
    out = {"dh": torch.empty(M, m, dtype=torch.float32, device=dev), "dx1": torch.empty(M, d, dtype=torch.float32, device=dev),
           "dgamma": torch.zeros(d, dtype=torch.float32, device=dev), "dbeta": torch.zeros(d, dtype=torch.float32, device=dev)}
    KA, dA, y_act, slope, alpha, w_adj, dA_out = 0, None, None, 0.0, 1.0, None, None
    if adjust is not None:
        dA, y_act, slope, alpha, w_adj = adjust
        dA, y_act, w_adj = f(dA), f(y_act), f(w_adj)
        KA = dA.shape[1]
        dx2 = torch.empty(M, d, dtype=torch.float32, device=dev)
        dA_out = torch.empty(M, KA, dtype=torch.float32, device=dev)
        out["dA"] = dA_out
    else:
        dx2 = f(dx2).clone()
    out["dx2"] = dx2
    w_proj, rs1, dO = None, None, None
    if proj is not None:
        w_proj, rs1 = f(proj[0]), f(proj[1])
        dO = out["dO"] = torch.empty(M, d, dtype=torch.float32, device=dev)
    keep = [f(hpre), f(x1), f(gamma), f(w_fc1), f(w_fc2), f(rs2)]
    nbytes = L.lib().srad_op_mlp_bwd_scratch_bytes(d, m, KA)
    sbuf, sp, sb = _scratch(nbytes, dev)
    L.check(L.lib().srad_op_mlp_bwd(M, d, m, L.dptr(dx2), L.dptr(keep[0]), L.dptr(keep[1]), L.dptr(keep[2]), L.dptr(keep[3]),
                                    L.dptr(keep[4]), L.dptr(keep[5]), int(rps), L.dptr(out["dh"]), L.dptr(out["dx1"]),
                                    L.dptr(out["dgamma"]), L.dptr(out["dbeta"]), KA, L.dptr(dA), KA, L.dptr(y_act), KA,
                                    float(slope), float(alpha), L.dptr(w_adj), L.dptr(dA_out), L.dptr(w_proj), L.dptr(rs1),
                                    L.dptr(dO), sp, sb, wgrad_workspace(dev), L.current_stream_ptr()), "op_mlp_bwd")
    return out


def lin_ln_bwd(dy: torch.Tensor, w: torch.Tensor, x: torch.Tensor, gamma: torch.Tensor, dres: Optional[torch.Tensor] = None,
               out: Optional[torch.Tensor] = None):
    """Data gradient of ``Linear(w [K, d])`` applied to ``LayerNorm(x)`` plus the LayerNorm backward, one launch
    (bf16 MFMA, ``srad_op_lin_ln_bwd``): returns (dres + LN'(dy @ w) (+ out if given), dgamma, dbeta)."""
    _need_cuda(dy, w, x, gamma)
    M, K = dy.shape
    d = w.shape[1]
    dev = x.device
    acc = out is not None
    if out is None:
        out = torch.empty(M, d, dtype=torch.float32, device=dev)
    dg = torch.zeros(d, dtype=torch.float32, device=dev)
    db = torch.zeros(d, dtype=torch.float32, device=dev)
    keep = [dy.detach().float().contiguous(), w.detach().float().contiguous(), gamma.detach().float().contiguous(),
            None if dres is None else dres.detach().float().contiguous()]
    nbytes = L.lib().srad_op_lin_ln_bwd_scratch_bytes(K, d)
    sbuf, sp, sb = _scratch(nbytes, dev)
    L.check(L.lib().srad_op_lin_ln_bwd(M, K, d, L.dptr(keep[0]), L.dptr(keep[1]), L.dptr(x), x.stride(0), L.dptr(keep[2]),
                                       L.dptr(keep[3]), L.dptr(out), out.stride(0), 1 if acc else 0, L.dptr(dg), L.dptr(db),
                                       sp, sb, wgrad_workspace(dev), L.current_stream_ptr()), "op_lin_ln_bwd")
    return out, dg, db


# ----------------------------------------------------------------------------------- C5: the 64 x 64-window attention's two kernels
def ln_qkv(x: torch.Tensor, ln_g: torch.Tensor, ln_b: torch.Tensor, w_qkv: torch.Tensor, b_qkv: torch.Tensor, heads: int,
           ws: int = 64, shift: int = 0) -> torch.Tensor:
    """norm1 + attn.qkv in one launch (``srad_op_ln_qkv``; src/drct.py:477, 278) -> the bf16 operands of the 64 x 64-window
    attention, [M, 3, heads, hdp]: q times head_dim^-0.5 log2 e, padding 0, column head_dim of the v slices 1."""
    _need_cuda(x, ln_g, ln_b, w_qkv, b_qkv)
    d = w_qkv.shape[1]
    assert x.dim() == 2 and x.stride(1) == 1 and x.dtype == torch.float32 and w_qkv.shape[0] == 3 * d
    qs = L.lib().srad_op_window_attn_qscale(ws, shift, d, heads)
    if qs == 0.0:
        raise ValueError(f"window {ws} / shift {shift} / head dim {d // heads}: not the bf16-operand attention path")
    hdp = (d // heads + 3) // 4 * 4
    f = lambda t: t.detach().float().contiguous()
    keep = [f(ln_g), f(ln_b), f(w_qkv), f(b_qkv)]
    out = torch.empty(x.shape[0], 3, heads, hdp, dtype=torch.bfloat16, device=x.device)
    sbuf, sp, sb = _scratch(L.lib().srad_op_ln_qkv_scratch_bytes(d, heads), x.device)
    L.check(L.lib().srad_op_ln_qkv(L.dptr(x), x.stride(0), x.shape[0], d, heads, L.dptr(keep[0]), L.dptr(keep[1]), L.dptr(keep[2]),
                                   L.dptr(keep[3]), L.dptr(out), hdp, float(qs), sp, sb, L.current_stream_ptr()), "op_ln_qkv")
    return out


def window_attention_bf16_in(qkv_h: torch.Tensor, table: torch.Tensor, B: int, H: int, W: int, ws: int, shift: int, d: int) -> torch.Tensor:
    """WindowAttention on ``ln_qkv``'s operands (``srad_op_window_attn_bf16_in``; src/drct.py:281-299, 482-504) -> [B*H*W, d] fp32."""
    _need_cuda(qkv_h, table)
    T, three, heads, hdp = qkv_h.shape
    assert three == 3 and T == B * H * W and qkv_h.dtype == torch.bfloat16 and qkv_h.is_contiguous()
    tb = table.detach().float().contiguous()
    out = torch.empty(T, d, dtype=torch.float32, device=qkv_h.device)
    L.check(L.lib().srad_op_window_attn_bf16_in(L.dptr(qkv_h), L.dptr(out), L.dptr(tb), B, H, W, ws, shift, d, heads, hdp,
                                                L.current_stream_ptr()), "op_window_attn_bf16_in")
    return out


# ----------------------------------------------------------------------------------- fused Swin-block halves (bf16, window 8)
def qkv_attn(x: torch.Tensor, ln_g: torch.Tensor, ln_b: torch.Tensor, w_qkv: torch.Tensor, b_qkv: torch.Tensor,
             table: torch.Tensor, B: int, H: int, W: int, shift: int, heads: int, out_bf16: bool = False,
             precision: str = "bf16") -> torch.Tensor:
    """First half of a Swin block in ONE launch (``srad_op_qkv_attn``; src/drct.py:477-504 up to attn.proj, 271-299):
    x [B*H*W, >=d] block input rows (columns [0, d) are read), w_qkv [3d, d], table [225, heads] -> attention output
    [B*H*W, d] in token order (window partition, cyclic shift and their inverses are index arithmetic inside); fp32, or
    with ``out_bf16`` the bf16 tensor the engines hand to ``mlp_block``.  precision "bf16" or "bf16x3" (split-bf16: fp32 out)."""
    _need_cuda(x, ln_g, ln_b, w_qkv, b_qkv, table)
    d = w_qkv.shape[1]
    assert x.dim() == 2 and x.shape[0] == B * H * W and x.stride(1) == 1 and x.dtype == torch.float32 and w_qkv.shape[0] == 3 * d
    f = lambda t: t.detach().float().contiguous()
    keep = [f(ln_g), f(ln_b), f(w_qkv), f(b_qkv), f(table)]
    out = torch.empty(x.shape[0], d, dtype=torch.bfloat16 if out_bf16 else torch.float32, device=x.device)
    sbuf, sp, sb = _scratch(L.lib().srad_op_swin_scratch_bytes(d, heads, 4, 4), x.device)
    L.check(L.lib().srad_op_qkv_attn(L.PRECISIONS[precision], L.dptr(x), x.stride(0), B, H, W, shift, d, heads, L.dptr(keep[0]), L.dptr(keep[1]),
                                     L.dptr(keep[2]), L.dptr(keep[3]), L.dptr(keep[4]), L.dptr(out), int(out_bf16), sp, sb,
                                     L.current_stream_ptr()), "op_qkv_attn")
    return out


def mlp_block(attn: torch.Tensor, shortcut: torch.Tensor, w_proj, b_proj, ln_g, ln_b, w_fc1, b_fc1, w_fc2, b_fc2, w_adj, b_adj, *,
              act: int = L.ACT_LRELU, slope: float = 0.2, alpha: float = 1.0, residual: Optional[torch.Tensor] = None,
              out: Optional[torch.Tensor] = None, out_offset: int = 0, fm: int = 0, precision: str = "bf16") -> torch.Tensor:
    """Second half of a Swin block + the RDG's 1x1 adjust conv in ONE launch (``srad_op_mlp_block``; src/drct.py:300,
    509-510, 184-190, 389-396).  attn [M, d] (crosses the boundary as bf16 - the operand the MFMA takes; an fp32 tensor is
    rounded here; with precision "bf16x3" it crosses as fp32), shortcut [M, >=d]; returns / fills out[:, out_offset:out_offset+no]."""
    _need_cuda(attn, shortcut, w_proj, w_fc1, w_fc2, w_adj)
    M, d = attn.shape
    m, no = w_fc1.shape[0], w_adj.shape[0]
    f = lambda t: t.detach().float().contiguous()
    keep = [f(w_proj), f(b_proj), f(ln_g), f(ln_b), f(w_fc1), f(b_fc1), f(w_fc2), f(b_fc2), f(w_adj.reshape(no, d)), f(b_adj)]
    attn = attn.detach().to(torch.float32 if L.PRECISIONS[precision] == L.PREC_BF16X3 else torch.bfloat16).contiguous()
    if out is None:
        out = torch.empty(M, no, dtype=torch.float32, device=attn.device)
    sbuf, sp, sb = _scratch(L.lib().srad_op_swin_scratch_bytes(d, 1, m, no), attn.device)
    L.check(L.lib().srad_op_mlp_block(L.PRECISIONS[precision], M, d, m, no, int(fm), L.dptr(attn), L.dptr(shortcut), shortcut.stride(0), *[L.dptr(k) for k in keep],
                                      int(act), float(slope), float(alpha), L.dptr(residual), 0 if residual is None else residual.stride(0),
                                      L.dptr(out), out.stride(0), int(out_offset), sp, sb, L.current_stream_ptr()), "op_mlp_block")
    return out
