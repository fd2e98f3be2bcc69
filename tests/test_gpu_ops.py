"""GPU: each hand-written kernel on its own against a torch fp32 statement of the same op
(SURVEY.md §8(a) rows A4-A7, A9, B3, B5).  Tolerances: fp32 mode 1e-4 relative (exact-fp32 MFMA,
only summation order differs); bf16 mode 3e-2 relative (8-bit mantissa inputs, fp32 accumulate)."""
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

TOL = {"fp32": 1e-4, "bf16": 3e-2, "bf16x3": 1e-4}     # bf16x3 = split-bf16 (hi + lo operands, three bf16 MFMAs per product)


def _rel(a, b):
    return float((a.double() - b.double()).abs().max() / b.double().abs().max().clamp_min(1e-30))


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    return torch.device("cuda:0")


@pytest.mark.parametrize("prec", ["fp32", "bf16", "bf16x3"])
@pytest.mark.parametrize("M,K,N", [(4096, 180, 540), (1000, 212, 32), (77, 308, 180), (256, 488, 244), (130, 36, 3)])
def test_linear_variants(dev, prec, M, K, N):
    from srad_amd import ops
    g = torch.Generator(device="cpu").manual_seed(M + K + N)
    x = torch.randn(M, K + 12, generator=g).to(dev)[:, :K]          # row stride != K
    w = (torch.randn(N, K, generator=g) / math.sqrt(K)).to(dev)
    b = torch.randn(N, generator=g).to(dev)
    r = torch.randn(M, N, generator=g).to(dev)
    y = ops.gemm(x, w, b, precision=prec)
    assert _rel(y, F.linear(x, w, b)) < TOL[prec]
    y = ops.gemm(x, w, b, act=1, precision=prec)                     # GELU
    assert _rel(y, F.gelu(F.linear(x, w, b))) < TOL[prec]
    y = ops.gemm(x, w, b, act=2, slope=0.2, alpha=0.2, residual=r, precision=prec)
    assert _rel(y, F.leaky_relu(F.linear(x, w, b), 0.2) * 0.2 + r) < TOL[prec]
    # LayerNorm prologue + offset store into a wider buffer
    lg, lb = torch.randn(K, generator=g).to(dev), torch.randn(K, generator=g).to(dev)
    out = torch.zeros(M, N + 40, device=dev)
    if K > 320 or K % 4:
        # the fused prologue keeps the whole row resident: up to 320 channels, multiple of 4
        with pytest.raises((RuntimeError, ValueError)):
            ops.gemm(x, w, b, ln=(lg, lb), out=out, out_offset=8, precision=prec)
        return
    ops.gemm(x, w, b, ln=(lg, lb), out=out, out_offset=8, precision=prec)
    ref = F.linear(F.layer_norm(x, (K,), lg, lb, 1e-5), w, b)
    assert _rel(out[:, 8:8 + N], ref) < TOL[prec]
    assert float(out[:, :8].abs().max()) == 0 and float(out[:, 8 + N:].abs().max()) == 0


@pytest.mark.parametrize("prec", ["fp32", "bf16", "bf16x3"])
@pytest.mark.parametrize("B,H,W,Cin,Cout,stride", [(2, 32, 32, 180, 64, 1), (1, 17, 23, 1, 180, 1), (2, 16, 12, 3, 20, 1),
                                                   (1, 32, 32, 20, 20, 2), (1, 15, 11, 40, 80, 2), (3, 8, 8, 64, 1, 1)])
def test_conv3x3(dev, prec, B, H, W, Cin, Cout, stride):
    from srad_amd import ops
    g = torch.Generator(device="cpu").manual_seed(B * H + Cin)
    x = torch.randn(B, Cin, H, W, generator=g).to(dev)
    w = (torch.randn(Cout, Cin, 3, 3, generator=g) / math.sqrt(9 * Cin)).to(dev)
    b = torch.randn(Cout, generator=g).to(dev)
    xn = x.permute(0, 2, 3, 1).reshape(-1, Cin).clone(memory_format=torch.contiguous_format)
    y = ops.gemm(xn, w, b, B=B, H=H, W=W, stride=stride, act=3, precision=prec)
    ref = F.relu(F.conv2d(x, w, b, stride=stride, padding=1))
    Ho, Wo = ref.shape[-2:]
    assert y.shape == (B * Ho * Wo, Cout)
    assert _rel(y, ref.permute(0, 2, 3, 1).reshape(-1, Cout)) < TOL[prec]


@pytest.mark.parametrize("B,H,W", [(2, 64, 64), (3, 64, 96), (8, 128, 64), (1, 128, 128)])
@pytest.mark.parametrize("mode", ["relu", "residual", "offset_out"])
def test_conv3x3_80_channels_weight_resident_kernel(dev, B, H, W, mode):
    """DRN-L's 80 -> 80 channel 3x3 convolution at sizes that take the weight-resident persistent kernel (bf16,
    kernels_conv80.hip; >= 8192 pixels, H % 4 == 0, W % 32 == 0): 128 - 512 tiles (one to two per workgroup), image borders,
    bias + ReLU, bias + residual, and writing into a column range of a wider output buffer - against torch's conv2d on
    bf16-rounded operands (fp32 accumulation on both sides: only the summation order differs)."""
    from srad_amd import ops
    g = torch.Generator(device="cpu").manual_seed(B * H + W)
    bf = lambda t: t.to(torch.bfloat16).float()
    x = bf(torch.randn(B, 80, H, W, generator=g)).to(dev)
    w = bf(torch.randn(80, 80, 3, 3, generator=g) / math.sqrt(9 * 80)).to(dev)
    b = torch.randn(80, generator=g).to(dev)
    xn = x.permute(0, 2, 3, 1).reshape(-1, 80).clone(memory_format=torch.contiguous_format)
    conv = F.conv2d(x, w, b, padding=1).permute(0, 2, 3, 1).reshape(-1, 80)
    if mode == "relu":
        y = ops.gemm(xn, w, b, B=B, H=H, W=W, act=3, precision="bf16")
        ref = F.relu(conv)
    elif mode == "residual":
        r = torch.randn(B * H * W, 80, generator=g).to(dev)
        y = ops.gemm(xn, w, b, B=B, H=H, W=W, residual=r, precision="bf16")
        ref = conv + r
    else:
        out = torch.full((B * H * W, 96), float("nan"), device=dev)
        ops.gemm(xn, w, b, B=B, H=H, W=W, out=out, out_offset=8, precision="bf16")
        assert torch.isnan(out[:, :8]).all() and torch.isnan(out[:, 88:]).all()
        y, ref = out[:, 8:88], conv
    assert _rel(y, ref) < 1e-4


@pytest.mark.parametrize("B,H,W", [(2, 64, 64), (8, 128, 128), (3, 64, 96)])
@pytest.mark.parametrize("mode", ["relu_h", "mask_h", "add_f32", "pool_h"])
def test_conv80_bf16_operands(dev, B, H, W, mode):
    """conv80_kernel<XH, RM> as DRN's bf16 chains issue it (round 3): bf16 input array, bf16 or fp32 output, the residual
    operand none / a bf16 ReLU mask (backward through the ReLU: out * (t > 0)) / an fp32 addend (the skip path of the data
    gradient), the per-tile column sums taken from the fp32 values - against torch's conv2d on the same bf16 operands.  A bf16
    output may differ from the rounded reference by one bf16 ulp (the summation order differs): 2^-7 relative."""
    from srad_amd import ops
    g = torch.Generator(device="cpu").manual_seed(7 * B + H + W)
    x_h = torch.randn(B * H * W, 80, generator=g).to(torch.bfloat16).to(dev)
    w = (torch.randn(80, 80, 3, 3, generator=g) / math.sqrt(720)).to(torch.bfloat16).float().to(dev)
    b = torch.randn(80, generator=g).to(dev)
    x = x_h.float().reshape(B, H, W, 80).permute(0, 3, 1, 2)
    conv = F.conv2d(x, w, b, padding=1).permute(0, 2, 3, 1).reshape(-1, 80)
    if mode == "relu_h":
        y = ops.conv80_bf16(x_h, w, b, B=B, H=H, W=W, act=3)
        ref = F.relu(conv)
    elif mode == "mask_h":
        t_h = torch.randn(B * H * W, 80, generator=g).to(torch.bfloat16).to(dev)
        t_h[::7] = 0                                                     # exact zeros: the mask is (t > 0), not (t >= 0)
        y = ops.conv80_bf16(x_h, w, None, B=B, H=H, W=W, residual=t_h, rmode=2, slope=0.0)
        ref = (conv - b) * (t_h.float() > 0)
    elif mode == "add_f32":
        r = torch.randn(B * H * W, 80, generator=g).to(dev)
        y = ops.conv80_bf16(x_h, w, None, B=B, H=H, W=W, residual=r, rmode=0, out_bf16=False)
        ref = conv - b + r
        assert y.dtype == torch.float32 and _rel(y, ref) < 1e-4
        return
    else:
        y, part = ops.conv80_bf16(x_h, w, b, B=B, H=H, W=W, pool=True)
        ref = conv
        tiles = conv.reshape(B, H // 4, 4, W // 32, 32, 80).sum((2, 4)).reshape(-1, 80)    # 4 x 32-pixel tiles, row-major per image
        assert _rel(part, tiles) < 1e-4
    assert y.dtype == torch.bfloat16
    assert float((y.float() - ref).abs().max()) <= 2.0 ** -7 * float(ref.abs().max())
    assert _rel(y.float(), ref) < 4e-3


@pytest.mark.parametrize("B,H,W,x_bf16", [(2, 64, 64, True), (8, 128, 128, True), (2, 64, 96, False)])
def test_wgrad_conv9_bf16_operands(dev, B, H, W, x_bf16):
    """wgrad_conv9_kernel<5, YH, XH> (round 3): the nine-tap weight gradient of the 80 -> 80 convolution from a bf16 dY and a bf16
    or fp32 X - the same numbers as the fp32-operand kernel on the bf16-rounded tensors (the MFMA operands are those bf16 values
    either way), and against torch."""
    from srad_amd import ops
    g = torch.Generator(device="cpu").manual_seed(B + H + W)
    dy_h = torch.randn(B * H * W, 80, generator=g).to(torch.bfloat16).to(dev)
    xx = torch.randn(B * H * W, 80, generator=g)
    x = (xx.to(torch.bfloat16) if x_bf16 else xx).to(dev)
    dw, db = ops.wgrad_conv9_bf16(dy_h, x, B=B, H=H, W=W)
    dw0, db0 = ops.wgrad(dy_h.float(), x.float().to(torch.bfloat16).float(), 80, 80, ntaps=9, B=B, H=H, W=W, precision="bf16")
    assert _rel(dw, dw0) < 1e-5 and _rel(db, db0) < 1e-5
    xt = x.float().to(torch.bfloat16).float().reshape(B, H, W, 80).permute(0, 3, 1, 2).requires_grad_(False)
    wt = torch.zeros(80, 80, 3, 3, device=dev, requires_grad=True)
    out = F.conv2d(xt, wt, None, padding=1)
    out.backward(dy_h.float().reshape(B, H, W, 80).permute(0, 3, 1, 2))
    assert _rel(dw.reshape(80, 80, 3, 3), wt.grad) < 1e-4
    assert _rel(db, dy_h.float().sum(0)) < 1e-5


@pytest.mark.parametrize("cin,n,mode", [(3, 20, "bias"), (40, 3, "bias"), (80, 1, "relu"), (4, 40, "residual"), (20, 3, "offset_out"), (4, 80, "bias"),
                                        (40, 3, "offset_out"), (80, 3, "residual"), (80, 4, "bias")])
def test_thin_convolutions(dev, cin, n, mode):
    """3x3 stride-1 convolutions with few input or output channels at >= 32768 pixels - DRN's head (3 -> 20), its tails (40 / 80 -> 3
    or 1) and their data gradients (4 -> 40 / 80).  Those with <= 8 INPUT channels take kernels_thin.hip in bf16 mode (round 3:
    direct fp32 FMAs, weights = the packed layer's bf16 values, activations fp32: 1e-5 against torch's conv2d on bf16-rounded
    weights); 40 / 80 -> <= 4 channels take `conv_tail_kernel` (MFMA with both operands from registers) and the rest the tiled GEMM
    (bf16-rounded activations either way: 4e-3).  Image borders, odd channel counts with
    element-wise stores, bias / ReLU / residual / column offset into a wider buffer."""
    from srad_amd import ops
    B, H, W = 2, 128, 160                                                  # 40960 pixels
    g = torch.Generator(device="cpu").manual_seed(cin * 100 + n)
    x = torch.randn(B, cin, H, W, generator=g).to(dev)
    w = (torch.randn(n, cin, 3, 3, generator=g) / math.sqrt(9 * cin)).to(torch.bfloat16).float().to(dev)
    b = torch.randn(n, generator=g).to(dev)
    xn = x.permute(0, 2, 3, 1).reshape(-1, cin).clone(memory_format=torch.contiguous_format)
    conv = F.conv2d(x, w, b, padding=1).permute(0, 2, 3, 1).reshape(-1, n)
    if mode == "bias":
        y, ref = ops.gemm(xn, w, b, B=B, H=H, W=W, precision="bf16"), conv
    elif mode == "relu":
        y, ref = ops.gemm(xn, w, b, B=B, H=H, W=W, act=3, precision="bf16"), F.relu(conv)
    elif mode == "residual":
        r = torch.randn(B * H * W, n, generator=g).to(dev)
        y, ref = ops.gemm(xn, w, b, B=B, H=H, W=W, residual=r, precision="bf16"), conv + r
    else:
        out = torch.full((B * H * W, 8), float("nan"), device=dev)
        ops.gemm(xn, w, b, B=B, H=H, W=W, out=out, out_offset=4, precision="bf16")
        assert torch.isnan(out[:, :4]).all() and torch.isnan(out[:, 4 + n:]).all()
        y, ref = out[:, 4:4 + n], conv
    assert _rel(y, ref) < (1e-5 if cin <= 8 else 4e-3), _rel(y, ref)


def test_conv80_eligible_shape_with_gelu_is_not_sent_to_the_weight_resident_kernel(dev):
    """conv80's epilogue implements none / ReLU / LeakyReLU; an 80 -> 80 3x3 convolution with another activation at a
    conv80-eligible shape must take the tiled GEMM (which applies GELU) instead of silently dropping the activation."""
    from srad_amd import ops
    g = torch.Generator(device="cpu").manual_seed(9)
    bf = lambda t: t.to(torch.bfloat16).float()
    B, H, W = 2, 64, 64
    x = bf(torch.randn(B, 80, H, W, generator=g)).to(dev)
    w = bf(torch.randn(80, 80, 3, 3, generator=g) / math.sqrt(720)).to(dev)
    b = torch.randn(80, generator=g).to(dev)
    xn = x.permute(0, 2, 3, 1).reshape(-1, 80).clone(memory_format=torch.contiguous_format)
    y = ops.gemm(xn, w, b, B=B, H=H, W=W, act=1, precision="bf16")
    ref = F.gelu(F.conv2d(x, w, b, padding=1)).permute(0, 2, 3, 1).reshape(-1, 80)
    assert _rel(y, ref) < 1e-3, _rel(y, ref)


@pytest.mark.parametrize("prec", ["fp32", "bf16", "bf16x3"])
def test_conv_pixel_shuffle(dev, prec):
    from srad_amd import ops
    g = torch.Generator(device="cpu").manual_seed(5)
    B, H, W, Cin = 2, 12, 10, 64
    x = torch.randn(B, Cin, H, W, generator=g).to(dev)
    w = (torch.randn(256, Cin, 3, 3, generator=g) / 24).to(dev)
    b = torch.randn(256, generator=g).to(dev)
    xn = x.permute(0, 2, 3, 1).reshape(-1, Cin).clone(memory_format=torch.contiguous_format)
    y = ops.gemm(xn, w, b, B=B, H=H, W=W, pixel_shuffle=True, precision=prec)
    ref = F.pixel_shuffle(F.conv2d(x, w, b, padding=1), 2).permute(0, 2, 3, 1).reshape(-1, 64)
    assert _rel(y, ref) < TOL[prec]


@pytest.mark.parametrize("B,H,W", [(2, 64, 64), (3, 64, 96), (2, 128, 128)])
def test_upsampler_conv_as_four_subpixel_convolutions(dev, B, H, W):
    """DRN's Upsampler convolution (80 -> 320 channels + PixelShuffle(2), src/drn.py:55-81) at sizes the weight-resident kernel takes
    (round 3: four 80 -> 80 launches, one per sub-pixel position - rows 4 c + q of the packed weight and bias, outputs at pixel
    (2 y + q / 2, 2 x + q % 2)): against torch's conv2d + pixel_shuffle on bf16-rounded operands, and against the tiled GEMM's
    pixel-shuffle epilogue (SRAD_NO_UPCONV_SPLIT is read per call)."""
    import os
    from srad_amd import ops
    g = torch.Generator(device="cpu").manual_seed(B + H + W)
    bf = lambda t: t.to(torch.bfloat16).float()
    x = bf(torch.randn(B, 80, H, W, generator=g)).to(dev)
    w = bf(torch.randn(320, 80, 3, 3, generator=g) / math.sqrt(720)).to(dev)
    b = torch.randn(320, generator=g).to(dev)
    xn = x.permute(0, 2, 3, 1).reshape(-1, 80).clone(memory_format=torch.contiguous_format)
    from srad_amd import _lib as L
    L.prof_enable(True)
    y = ops.gemm(xn, w, b, B=B, H=H, W=W, pixel_shuffle=True, precision="bf16")
    torch.cuda.synchronize()
    prof = L.prof_collect()
    os.environ["SRAD_NO_UPCONV_SPLIT"] = "1"
    try:
        y0 = ops.gemm(xn, w, b, B=B, H=H, W=W, pixel_shuffle=True, precision="bf16")
        torch.cuda.synchronize()
        prof0 = L.prof_collect()
    finally:
        del os.environ["SRAD_NO_UPCONV_SPLIT"]
        L.prof_enable(False)
    assert prof.get("conv80", {}).get("launches") == 4 and "conv80" not in prof0, (prof, prof0)     # the two paths were taken
    ref = F.pixel_shuffle(F.conv2d(x, w, b, padding=1), 2).permute(0, 2, 3, 1).reshape(-1, 80)
    assert y.shape == ref.shape and _rel(y, ref) < 1e-4, _rel(y, ref)
    assert _rel(y0, ref) < 1e-4 and _rel(y, y0) < 1e-4


def _attn_ref(qkv, table, B, H, W, ws, shift, heads):
    from oracle import sr_ref as R
    d = qkv.shape[1] // 3
    N = ws * ws
    x = qkv.view(B, H, W, 3 * d)
    if shift:
        x = torch.roll(x, (-shift, -shift), (1, 2))
    xw = R.window_partition(x, ws).view(-1, N, 3, heads, d // heads).permute(2, 0, 3, 1, 4)
    q, k, v = xw[0] * (d // heads) ** -0.5, xw[1], xw[2]
    a = q @ k.transpose(-2, -1)
    bias = table[R.rel_pos_index(ws).view(-1)].view(N, N, -1).permute(2, 0, 1)
    a = a + bias.unsqueeze(0)
    if shift:
        m = R.calculate_mask(H, W, ws, shift)
        nW = m.shape[0]
        a = (a.view(B, nW, heads, N, N) + m.unsqueeze(1).unsqueeze(0)).view(-1, heads, N, N)
    o = (torch.softmax(a, -1) @ v).transpose(1, 2).reshape(-1, ws, ws, d)
    o = R.window_reverse(o, ws, H, W)
    if shift:
        o = torch.roll(o, (shift, shift), (1, 2))
    return o.reshape(B * H * W, d)


@pytest.mark.parametrize("prec", ["fp32", "bf16", "bf16x3"])
@pytest.mark.parametrize("B,H,W,ws,shift,d,heads", [
    (2, 32, 32, 8, 0, 180, 6), (2, 32, 32, 8, 4, 212, 4), (1, 32, 32, 8, 0, 244, 2), (1, 32, 32, 8, 4, 276, 6),
    (1, 32, 32, 8, 0, 308, 4), (1, 64, 32, 8, 4, 212, 4), (2, 16, 16, 4, 2, 180, 6), (1, 8, 8, 2, 1, 212, 4),
    (1, 64, 64, 16, 8, 276, 6), (1, 48, 24, 24, 12, 60, 2)])
def test_window_attention(dev, prec, B, H, W, ws, shift, d, heads):
    from srad_amd import ops
    g = torch.Generator(device="cpu").manual_seed(ws * 100 + d)
    qkv = torch.randn(B * H * W, 3 * d, generator=g)
    qkv[:, :2 * d] *= 1.5                                           # make softmax peaky
    table = torch.randn((2 * ws - 1) ** 2, heads, generator=g)
    ref = _attn_ref(qkv, table, B, H, W, ws, shift, heads)
    out = ops.window_attention(qkv.to(dev), table.to(dev), B, H, W, ws, shift, heads, precision=prec)
    assert _rel(out.cpu(), ref) < TOL[prec]


def test_layernorm(dev):
    from srad_amd import ops
    g = torch.Generator(device="cpu").manual_seed(1)
    x = (torch.randn(1000, 308, generator=g) * 3 + 1).to(dev)[:, :180]
    w, b = torch.randn(180, generator=g).to(dev), torch.randn(180, generator=g).to(dev)
    assert _rel(ops.layernorm(x, w, b), F.layer_norm(x, (180,), w, b, 1e-5)) < 1e-5


def test_bad_arguments_raise(dev):
    from srad_amd import ops
    qkv = torch.zeros(30 * 30, 3 * 12, device=dev)
    with pytest.raises(RuntimeError, match="multiple of window"):
        ops.window_attention(qkv, torch.zeros(225, 2, device=dev), 1, 30, 30, 8, 0, 2)
    with pytest.raises(RuntimeError, match="GPU only"):
        ops.layernorm(torch.zeros(4, 8), torch.zeros(8), torch.zeros(8))


def test_window_attention_ignores_pad_columns(dev):
    """Head slices are padded to a multiple of 4 floats; the pad columns may hold anything (NaN here): selects, not
    multiplications by zero, keep them out of q k^T and P V."""
    from srad_amd import ops
    g = torch.Generator(device="cpu").manual_seed(5)
    B, H, W, ws, d, heads = 1, 16, 16, 8, 180, 6           # head_dim 30 -> padded to 32
    qkv = torch.randn(B * H * W, 3 * d, generator=g).to(dev)
    table = torch.randn((2 * ws - 1) ** 2, heads, generator=g).to(dev)
    for prec in ("fp32", "bf16"):
        a = ops.window_attention(qkv, table, B, H, W, ws, 4, heads, precision=prec)
        b = ops.window_attention(qkv, table, B, H, W, ws, 4, heads, precision=prec, pad_value=float("nan"))
        assert bool(torch.isfinite(b).all()) and torch.equal(a, b)


def test_tail_convolution_at_the_c3_shape(dev):
    """DRN-L's last tail (40 -> 3 channels at 256 px x 8 images, src/drn.py:265-267) on `conv_tail_kernel`'s 512-workgroup
    instance (16 tiles per wave): against torch's conv2d on bf16-rounded operands at 1e-5 (only the summation order differs)."""
    from srad_amd import ops
    B, H, W, cin, n = 8, 256, 256, 40, 3
    g = torch.Generator(device="cpu").manual_seed(77)
    bf = lambda t: t.to(torch.bfloat16).float()
    x = bf(torch.randn(B, cin, H, W, generator=g)).to(dev)
    w = bf(torch.randn(n, cin, 3, 3, generator=g) / math.sqrt(9 * cin)).to(dev)
    b = torch.randn(n, generator=g).to(dev)
    xn = x.permute(0, 2, 3, 1).reshape(-1, cin).clone(memory_format=torch.contiguous_format)
    y = ops.gemm(xn, w, b, B=B, H=H, W=W, precision="bf16")
    ref = F.conv2d(x, w, b, padding=1).permute(0, 2, 3, 1).reshape(-1, n)
    assert _rel(y, ref) < 1e-5, _rel(y, ref)
