"""GPU: the two fused bf16 launches a Swin block becomes in the benched configuration (qkv_attn_kernel, mlp_block_kernel -
120 of the 132 launches of the C2 forward), each on its own against the oracle, through the C ABI (srad_op_qkv_attn,
srad_op_mlp_block).

Reference: oracle.sr_ref.swin_block (/root/reference/src/drct.py:472-512) in fp32 on bf16-ROUNDED weights, with the
oracle's ``rnd`` hook rounding exactly the tensors the kernels round when they stage MFMA operands (LayerNorm outputs, q k v,
the un-normalised probabilities, GELU(fc1), x2).  What is left is accumulation order, __expf / rsqrt / the 1.5e-7 erf
polynomial, and bf16 roundings that flip because of those: a wrong relative-position-bias entry, a dropped shift-mask region
or a mis-indexed head would be O(1e-1).  Bar 2e-3 of the output's max (measured values are printed).

Coverage: the five (dim, heads) rows of SURVEY.md §8's block table, shift 0 and 4, image geometries with one and several
windows per side (incl. non-square), and for the MLP half every tile variant (16 / 32 / 64 rows per workgroup) at 4096 / 8192
/ 65536-token-style row counts (scaled down to rows that keep the CPU oracle in seconds: the tile variant is what is selected)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import sr_ref as R
from srad_amd import _lib as L
from srad_amd import ops

pytestmark = pytest.mark.gpu

BLOCKS = [(180, 6, 360), (212, 4, 424), (244, 2, 488), (276, 6, 276), (308, 4, 308)]     # (d, heads, mlp hidden): SURVEY.md §8
BAR = 2e-3


def bf16r(t: torch.Tensor) -> torch.Tensor:
    return t.bfloat16().float()


def make_block(d, heads, hidden, no, seed, round_weights=True):
    g = torch.Generator().manual_seed(seed)
    rn = lambda *s, std=1.0: torch.randn(*s, generator=g) * std
    bf16r = (lambda t: t.bfloat16().float()) if round_weights else (lambda t: t)
    sd = {"norm1.weight": 1 + 0.2 * rn(d), "norm1.bias": 0.1 * rn(d),
          "attn.qkv.weight": bf16r(rn(3 * d, d, std=d ** -0.5)), "attn.qkv.bias": 0.2 * rn(3 * d),
          "attn.relative_position_bias_table": rn(225, heads, std=0.5),
          "attn.proj.weight": bf16r(rn(d, d, std=d ** -0.5)), "attn.proj.bias": 0.1 * rn(d),
          "norm2.weight": 1 + 0.2 * rn(d), "norm2.bias": 0.1 * rn(d),
          "mlp.fc1.weight": bf16r(rn(hidden, d, std=d ** -0.5)), "mlp.fc1.bias": 0.1 * rn(hidden),
          "mlp.fc2.weight": bf16r(rn(d, hidden, std=hidden ** -0.5)), "mlp.fc2.bias": 0.1 * rn(d),
          "adjust.weight": bf16r(rn(no, d, std=d ** -0.5)), "adjust.bias": 0.1 * rn(no)}
    return sd


def rel(a, b):
    return float((a - b).abs().max() / b.abs().max())


@pytest.mark.parametrize("d,heads,hidden", BLOCKS)
@pytest.mark.parametrize("shift", [0, 4])
@pytest.mark.parametrize("B,H,W", [(2, 16, 24), (1, 8, 8), (4, 32, 32)])
def test_qkv_attn_kernel_matches_oracle(d, heads, hidden, shift, B, H, W):
    if (B, H, W) == (4, 32, 32) and (d, shift) not in ((180, 4), (308, 0), (244, 4)):
        pytest.skip("the C2 geometry is run on three block shapes (CPU oracle time)")
    sd = make_block(d, heads, hidden, 32, seed=d + shift)
    g = torch.Generator().manual_seed(7)
    D = 308                                                  # rows of the RDG's dense buffer: the block reads columns [0, d)
    x = torch.randn(B, H * W, D, generator=g) * 1.5 + 0.3
    taps = {}
    R.swin_block({k: v for k, v in sd.items()}, "", x[..., :d].contiguous(), H, W, 8, heads, shift, rnd=bf16r, taps=taps)
    ref = taps["attn"].reshape(B * H * W, d)
    xg = x.reshape(B * H * W, D).cuda()
    out = ops.qkv_attn(xg, sd["norm1.weight"].cuda(), sd["norm1.bias"].cuda(), sd["attn.qkv.weight"].cuda(), sd["attn.qkv.bias"].cuda(),
                       sd["attn.relative_position_bias_table"].cuda(), B, H, W, shift, heads)
    e = rel(out.cpu(), ref)
    # sensitivity of this check: one bias-table entry off by 0.5 or the mask dropped must exceed the bar by far
    print(f"qkv_attn d={d} heads={heads} shift={shift} {B}x{H}x{W}: rel err {e:.2e}")
    assert e < BAR, e
    assert not torch.isnan(out).any()
    # the bf16 hand-off the engines use (qkv_attn -> mlp_block) is the same values, rounded once
    out_h = ops.qkv_attn(xg, sd["norm1.weight"].cuda(), sd["norm1.bias"].cuda(), sd["attn.qkv.weight"].cuda(), sd["attn.qkv.bias"].cuda(),
                         sd["attn.relative_position_bias_table"].cuda(), B, H, W, shift, heads, out_bf16=True)
    assert out_h.dtype == torch.bfloat16 and torch.equal(out_h, out.to(torch.bfloat16))


def test_qkv_attn_check_is_sensitive_to_bias_and_mask():
    """The same comparison fails clearly for the bugs the whole-model bars of round 1 could not see."""
    d, heads, hidden, B, H, W, shift = 212, 4, 424, 1, 16, 16, 4
    sd = make_block(d, heads, hidden, 32, seed=1)
    x = torch.randn(B, H * W, d, generator=torch.Generator().manual_seed(3))
    args = lambda s: (x.reshape(-1, d).cuda(), s["norm1.weight"].cuda(), s["norm1.bias"].cuda(), s["attn.qkv.weight"].cuda(),
                      s["attn.qkv.bias"].cuda(), s["attn.relative_position_bias_table"].cuda(), B, H, W)
    good = ops.qkv_attn(*args(sd), shift, heads).cpu()
    taps = {}
    R.swin_block(sd, "", x, H, W, 8, heads, shift, rnd=bf16r, taps=taps)
    assert rel(good, taps["attn"].reshape(-1, d)) < BAR
    bad = dict(sd)
    bad["attn.relative_position_bias_table"] = sd["attn.relative_position_bias_table"].clone()
    bad["attn.relative_position_bias_table"][37, 2] += 0.5                       # ONE entry of one head
    assert rel(ops.qkv_attn(*args(bad), shift, heads).cpu(), taps["attn"].reshape(-1, d)) > 5 * BAR
    assert rel(ops.qkv_attn(*args(sd), 0, heads).cpu(), taps["attn"].reshape(-1, d)) > 20 * BAR      # shift / mask ignored


@pytest.mark.parametrize("d,heads,hidden", BLOCKS)
@pytest.mark.parametrize("fm,M", [(16, 4096), (32, 8192), (64, 2048), (0, 512)])
def test_mlp_block_kernel_matches_oracle(d, heads, hidden, fm, M):
    """fm 16 / 32 / 64 = the tile variants the engine picks at 4096 (C2) / 8192 (C4) / >= 32768 (C5) tokens.  Blocks 1-4 end
    in adjust_k (32 outputs + LeakyReLU(0.2), written in place behind the block's own columns); block 5 (d = 308) ends in
    adjust5 (180 outputs, * 0.2 + x, into the next RDG buffer)."""
    last = d == 308
    no = 180 if last else 32
    sd = make_block(d, heads, hidden, no, seed=3 * d + fm)
    g = torch.Generator().manual_seed(11)
    D = 308
    dense = torch.randn(M, D, generator=g) * 1.2
    attn = bf16r(torch.randn(M, d, generator=g) * 0.8)        # the kernel rounds its A operand to bf16: give both sides that
    # oracle: second half of swin_block, then the adjust conv (rdg, src/drct.py:389-396)
    x1 = dense[:, :d] + F.linear(attn, sd["attn.proj.weight"], sd["attn.proj.bias"])
    y = bf16r(F.layer_norm(x1, (d,), sd["norm2.weight"], sd["norm2.bias"], 1e-5))
    y = bf16r(F.gelu(F.linear(y, sd["mlp.fc1.weight"], sd["mlp.fc1.bias"])))
    x2 = x1 + F.linear(y, sd["mlp.fc2.weight"], sd["mlp.fc2.bias"])
    a = F.linear(bf16r(x2), sd["adjust.weight"], sd["adjust.bias"])
    ref = (a * 0.2 + dense[:, :180]) if last else F.leaky_relu(a, 0.2)
    dg = dense.cuda()
    c = lambda k: sd[k].cuda()
    if last:
        out = torch.full((M, D), float("nan"), device="cuda")
        ops.mlp_block(attn.cuda(), dg, c("attn.proj.weight"), c("attn.proj.bias"), c("norm2.weight"), c("norm2.bias"), c("mlp.fc1.weight"),
                      c("mlp.fc1.bias"), c("mlp.fc2.weight"), c("mlp.fc2.bias"), c("adjust.weight"), c("adjust.bias"), act=L.ACT_NONE,
                      slope=0.0, alpha=0.2, residual=dg, out=out, out_offset=0, fm=fm)
        got = out[:, :180].cpu()
        assert torch.isnan(out[:, 180:]).all()               # nothing written outside the adjust output's columns
    else:
        before = dg.clone()
        ops.mlp_block(attn.cuda(), dg, c("attn.proj.weight"), c("attn.proj.bias"), c("norm2.weight"), c("norm2.bias"), c("mlp.fc1.weight"),
                      c("mlp.fc1.bias"), c("mlp.fc2.weight"), c("mlp.fc2.bias"), c("adjust.weight"), c("adjust.bias"), act=L.ACT_LRELU,
                      slope=0.2, alpha=1.0, out=dg, out_offset=d, fm=fm)      # in place: torch.cat((x, x_k), -1) without a copy
        got = dg[:, d:d + 32].cpu()
        assert torch.equal(dg[:, :d], before[:, :d]) and torch.equal(dg[:, d + 32:], before[:, d + 32:])
    e = rel(got, ref)
    print(f"mlp_block d={d} m={hidden} no={no} fm={fm} M={M}: rel err {e:.2e}")
    assert e < BAR, e


# ------------------------------------------------------------------------------------------------------------------------
# split-bf16 ("bf16x3": every MFMA operand as hi + lo bf16 terms, three MFMAs per product): the SAME two kernels against the
# plain fp32 oracle - no rounding hook, weights NOT pre-rounded.  Bar 2e-4 of the output's max.
# ------------------------------------------------------------------------------------------------------------------------
BAR_X3 = 2e-4


@pytest.mark.parametrize("d,heads,hidden", BLOCKS)
@pytest.mark.parametrize("shift", [0, 4])
@pytest.mark.parametrize("B,H,W", [(2, 16, 24), (1, 8, 8), (4, 32, 32)])
def test_qkv_attn_split_bf16_matches_fp32_oracle(d, heads, hidden, shift, B, H, W):
    if (B, H, W) == (4, 32, 32) and (d, shift) not in ((180, 4), (308, 0), (244, 4), (276, 4)):
        pytest.skip("the C2 geometry is run on four block shapes (CPU oracle time)")
    sd = make_block(d, heads, hidden, 32, seed=d + shift, round_weights=False)
    g = torch.Generator().manual_seed(7)
    D = 308
    x = torch.randn(B, H * W, D, generator=g) * 1.5 + 0.3
    taps = {}
    R.swin_block({k: v for k, v in sd.items()}, "", x[..., :d].contiguous(), H, W, 8, heads, shift, taps=taps)      # fp32, no rnd hook
    ref = taps["attn"].reshape(B * H * W, d)
    out = ops.qkv_attn(x.reshape(B * H * W, D).cuda(), sd["norm1.weight"].cuda(), sd["norm1.bias"].cuda(), sd["attn.qkv.weight"].cuda(),
                       sd["attn.qkv.bias"].cuda(), sd["attn.relative_position_bias_table"].cuda(), B, H, W, shift, heads, precision="bf16x3")
    e = rel(out.cpu(), ref)
    print(f"qkv_attn bf16x3 d={d} heads={heads} shift={shift} {B}x{H}x{W}: rel err {e:.2e}")
    assert out.dtype == torch.float32 and not torch.isnan(out).any()
    assert e < BAR_X3, e
    # the plain bf16 kernel on the same unrounded operands is two orders of magnitude off that bar (what the mode is for)
    e16 = rel(ops.qkv_attn(x.reshape(B * H * W, D).cuda(), sd["norm1.weight"].cuda(), sd["norm1.bias"].cuda(), sd["attn.qkv.weight"].cuda(),
                           sd["attn.qkv.bias"].cuda(), sd["attn.relative_position_bias_table"].cuda(), B, H, W, shift, heads).cpu(), ref)
    assert e16 > 5 * BAR_X3, e16


@pytest.mark.parametrize("d,heads,hidden", BLOCKS)
@pytest.mark.parametrize("fm,M", [(16, 4096), (32, 8192), (0, 512), (0, 16384)])
def test_mlp_block_split_bf16_matches_fp32_oracle(d, heads, hidden, fm, M):
    last = d == 308
    no = 180 if last else 32
    sd = make_block(d, heads, hidden, no, seed=3 * d + fm, round_weights=False)
    g = torch.Generator().manual_seed(11)
    D = 308
    dense = torch.randn(M, D, generator=g) * 1.2
    attn = torch.randn(M, d, generator=g) * 0.8                  # fp32 hand-off, not rounded
    x1 = dense[:, :d] + F.linear(attn, sd["attn.proj.weight"], sd["attn.proj.bias"])
    y = F.layer_norm(x1, (d,), sd["norm2.weight"], sd["norm2.bias"], 1e-5)
    y = F.gelu(F.linear(y, sd["mlp.fc1.weight"], sd["mlp.fc1.bias"]))
    x2 = x1 + F.linear(y, sd["mlp.fc2.weight"], sd["mlp.fc2.bias"])
    a = F.linear(x2, sd["adjust.weight"], sd["adjust.bias"])
    ref = (a * 0.2 + dense[:, :180]) if last else F.leaky_relu(a, 0.2)
    dg = dense.cuda()
    c = lambda k: sd[k].cuda()
    args = (c("attn.proj.weight"), c("attn.proj.bias"), c("norm2.weight"), c("norm2.bias"), c("mlp.fc1.weight"), c("mlp.fc1.bias"),
            c("mlp.fc2.weight"), c("mlp.fc2.bias"), c("adjust.weight"), c("adjust.bias"))
    if last:
        out = torch.full((M, D), float("nan"), device="cuda")
        ops.mlp_block(attn.cuda(), dg, *args, act=L.ACT_NONE, slope=0.0, alpha=0.2, residual=dg, out=out, out_offset=0, fm=fm, precision="bf16x3")
        got = out[:, :180].cpu()
        assert torch.isnan(out[:, 180:]).all()
    else:
        before = dg.clone()
        ops.mlp_block(attn.cuda(), dg, *args, act=L.ACT_LRELU, slope=0.2, alpha=1.0, out=dg, out_offset=d, fm=fm, precision="bf16x3")
        got = dg[:, d:d + 32].cpu()
        assert torch.equal(dg[:, :d], before[:, :d]) and torch.equal(dg[:, d + 32:], before[:, d + 32:])
    e = rel(got, ref)
    print(f"mlp_block bf16x3 d={d} m={hidden} no={no} fm={fm} M={M}: rel err {e:.2e}")
    assert e < BAR_X3, e


def test_split_bf16_ops_refuse_what_they_do_not_do():
    x = torch.zeros(64, 180, device="cuda")
    w = torch.zeros(540, 180, device="cuda")
    with pytest.raises(RuntimeError, match="writes fp32"):
        ops.qkv_attn(x, x[0], x[0], w, w[:, 0], torch.zeros(225, 6, device="cuda"), 1, 8, 8, 0, 6, out_bf16=True, precision="bf16x3")
    with pytest.raises(RuntimeError, match="16 or 32 rows"):
        ops.mlp_block(x, x, w[:180], w[0], w[0], w[0], w[:360], torch.zeros(360, device="cuda"), torch.zeros(180, 360, device="cuda"), w[0],
                      w[:32], torch.zeros(32, device="cuda"), fm=64, precision="bf16x3")


def test_fused_ops_report_bad_arguments():
    x = torch.zeros(64, 180, device="cuda")
    w = torch.zeros(540, 180, device="cuda")
    with pytest.raises(RuntimeError, match="unsupported shape"):
        ops.qkv_attn(x, x[0], x[0], w, w[:, 0], torch.zeros(225, 5, device="cuda"), 1, 8, 8, 0, 5)       # 180 % 5 == 0 but no instance
    with pytest.raises(RuntimeError, match="GPU only"):
        ops.qkv_attn(x.cpu(), x[0], x[0], w, w[:, 0], torch.zeros(225, 6), 1, 8, 8, 0, 6)
