"""GPU: the two kernels that are 41 of the 47 ms of BASELINE config C5 (DRCT-L eval, 1024 px tile, window_size =
img_size // 4 = 64, src/main.py:286), each on its own through the C ABI against the oracle:

  * ``ln_qkv_kernel`` (srad_op_ln_qkv): norm1 + attn.qkv (src/drct.py:477, 278) -> the bf16 operands of the attention
    (q pre-scaled by head_dim^-0.5 log2 e, zero padding, V's ones column - those checked bit for bit);
  * ``window_attn_kernel`` ROW64 / QH path (srad_op_window_attn_bf16_in): 4096-token windows, online softmax over 64-key chunks
    that are rows of the window, the relative-position bias rows sliding through a four-slot LDS ring, the 0 / -100 shift mask,
    the softmax denominator out of P.V's ones column (src/drct.py:271-302, 449-470, 482-504).

Reference: ``oracle.sr_ref.attention_from_qkv`` / ``attention_core`` with the ``rnd`` hook (bf16 roundings of q k v and the probabilities),
``q_fold = log2 e`` (the kernel rounds q AFTER folding log2 e into it) and ``online_chunk = 64`` (probabilities rounded relative to
the running maximum of a streaming softmax, denominator summed from the rounded probabilities - without it the two sides round a
peaky row's probabilities at different scales and differ by 2 - 4e-3, measured), window by window so the 4096 x 4096 score tiles
stay small.
All five (dim, heads) rows of SURVEY.md §8's block table, shift 0 and 32, a 128 x 128 token image = two windows per side, so every
region pair of the shift mask occurs and so does an unmasked window.  Bar 2e-3 of the output's max, measured values printed."""
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import sr_ref as R
from srad_amd import ops

pytestmark = pytest.mark.gpu

BLOCKS = [(180, 6), (212, 4), (244, 2), (276, 6), (308, 4)]
BAR = 2e-3
LOG2E = 1.4426950408889634


def bf16r(t):
    return t.bfloat16().float()


def make(d, heads, seed, table_std=0.5):
    g = torch.Generator().manual_seed(seed)
    rn = lambda *s, std=1.0: torch.randn(*s, generator=g) * std
    return {"norm1.weight": 1 + 0.2 * rn(d), "norm1.bias": 0.1 * rn(d),
            "qkv.weight": bf16r(rn(3 * d, d, std=1.6 * d ** -0.5)), "qkv.bias": 0.2 * rn(3 * d),
            "relative_position_bias_table": rn(127 * 127, heads, std=table_std)}


def oracle_attention(qkv_h, table, H, W, ws, heads, d, shift, qscale_fold):
    """roll + partition + ``attention_from_qkv`` + reverse + roll back (src/drct.py:482-504, 282-299), one window at a time, on the
    attention kernel's OWN operands: qkv_h [T, 3, heads, hdp] bf16 as ln_qkv wrote them (q carries head_dim^-0.5 log2 e)."""
    hd = d // heads
    t = qkv_h.float()[..., :hd]                                     # [T, 3, heads, hd]; V's ones column and the padding dropped
    t = t.reshape(1, H, W, 3 * heads * hd)
    if shift:
        t = torch.roll(t, (-shift, -shift), (1, 2))
    tw = R.window_partition(t, ws).view(-1, ws * ws, 3, heads, hd).permute(2, 0, 3, 1, 4)     # [3, nW, heads, N, hd]
    q, k, v = tw[0] / LOG2E, tw[1], tw[2]                           # q back to the reference's scale; rnd(q * q_fold) restores it bit for bit
    mask = R.calculate_mask(H, W, ws, shift) if shift else None
    outs = []
    for i in range(q.shape[0]):
        outs.append(R.attention_from_qkv(q[i:i + 1], k[i:i + 1], v[i:i + 1], table, ws, None if mask is None else mask[i:i + 1],
                                         rnd=bf16r, q_fold=LOG2E, online_chunk=ws))      # the kernel's 64-key chunks = rows of the window
    o = torch.cat(outs).transpose(1, 2).reshape(-1, ws, ws, d)
    o = R.window_reverse(o, ws, H, W)
    if shift:
        o = torch.roll(o, (shift, shift), (1, 2))
    return o.reshape(H * W, d)


@pytest.mark.parametrize("d,heads", BLOCKS)
def test_ln_qkv_kernel_matches_oracle(d, heads):
    hd = d // heads
    hdp = (hd + 3) // 4 * 4
    sd = make(d, heads, seed=d)
    g = torch.Generator().manual_seed(3)
    M, D = 1024, 308
    x = torch.randn(M, D, generator=g) * 1.5 + 0.3
    qs = hd ** -0.5 * LOG2E
    xn = bf16r(F.layer_norm(x[:, :d], (d,), sd["norm1.weight"], sd["norm1.bias"], 1e-5))
    qkv = F.linear(xn, sd["qkv.weight"], sd["qkv.bias"]).view(M, 3, heads, hd)
    qkv[:, 0] *= qs
    got = ops.ln_qkv(x.cuda(), sd["norm1.weight"].cuda(), sd["norm1.bias"].cuda(), sd["qkv.weight"].cuda(), sd["qkv.bias"].cuda(), heads).cpu()
    assert got.shape == (M, 3, heads, hdp) and got.dtype == torch.bfloat16
    # the padding the attention kernel relies on, bit for bit: zeros, and 1 in column head_dim of the v slices
    pad = got[..., hd:].float()
    want = torch.zeros_like(pad)
    want[:, 2, :, 0] = 1.0
    assert torch.equal(pad, want)
    # the values: bf16 roundings of sums that differ in accumulation order (and in the odd bf16 flip of a LayerNorm output): the
    # kernel's value is the oracle's own rounding nearly everywhere, and never further off than one bf16 ulp of the largest value
    a, b = got[..., :hd].float(), qkv
    flips = float((a != bf16r(b)).float().mean())
    e = float((a - b).abs().max() / b.abs().max())
    print(f"ln_qkv d={d} heads={heads}: max err / max {e:.2e}, {100 * flips:.2f} % of the bf16 values one ulp off the oracle's rounding")
    assert e < 4e-3 and flips < 0.05


@pytest.mark.parametrize("d,heads", BLOCKS)
@pytest.mark.parametrize("shift", [0, 32])
def test_window64_attention_kernel_matches_oracle(d, heads, shift):
    ws, H, W = 64, 128, 128
    sd = make(d, heads, seed=7 * d + shift)
    g = torch.Generator().manual_seed(11 + shift)
    x = torch.randn(H * W, d, generator=g) * 1.5 + 0.3
    qkv_h = ops.ln_qkv(x.cuda(), sd["norm1.weight"].cuda(), sd["norm1.bias"].cuda(), sd["qkv.weight"].cuda(), sd["qkv.bias"].cuda(), heads,
                       ws=ws, shift=shift)
    out = ops.window_attention_bf16_in(qkv_h, sd["relative_position_bias_table"].cuda(), 1, H, W, ws, shift, d).cpu()
    with torch.no_grad():
        ref = oracle_attention(qkv_h.cpu(), sd["relative_position_bias_table"], H, W, ws, heads, d, shift, LOG2E)
    e = float((out - ref).abs().max() / ref.abs().max())
    # end to end (LayerNorm1 + qkv + attention against the oracle from x): the odd bf16 flip of a q / k element moves a peaky
    # row's probabilities, so this one is looser - it pins the pair, the line above pins the attention kernel
    with torch.no_grad():
        xn = bf16r(F.layer_norm(x, (d,), sd["norm1.weight"], sd["norm1.bias"], 1e-5)).view(1, H, W, d)
        if shift:
            xn = torch.roll(xn, (-shift, -shift), (1, 2))
        xw = R.window_partition(xn, ws).view(-1, ws * ws, d)
        mask = R.calculate_mask(H, W, ws, shift) if shift else None
        o = torch.cat([R.attention_core(sd, "", xw[i:i + 1], ws, heads, None if mask is None else mask[i:i + 1], rnd=bf16r, q_fold=LOG2E,
                                        online_chunk=ws) for i in range(xw.shape[0])])
        o = R.window_reverse(o.view(-1, ws, ws, d), ws, H, W)
        if shift:
            o = torch.roll(o, (shift, shift), (1, 2))
        e2e = float((out - o.reshape(H * W, d)).abs().max() / o.abs().max())
    print(f"window-64 attention d={d} heads={heads} shift={shift}: rel err {e:.2e} on the kernel's operands, {e2e:.2e} end to end from x")
    assert not torch.isnan(out).any()
    assert e < BAR, e
    assert e2e < 6e-3, e2e


@pytest.mark.parametrize("d,heads", [(180, 6), (244, 2), (308, 4)])
@pytest.mark.parametrize("shift", [0, 32])
def test_window64_attention_split_bf16_matches_the_plain_fp32_oracle(d, heads, shift):
    """The split-bf16 instance of the window attention kernel (round 3: Q, K, V tiles as hi + lo bf16 planes, probabilities split in
    registers, three bf16 MFMAs per product, the one-row-per-chunk path with the bias ring) at 64 x 64 windows against the PLAIN
    fp32 oracle - no rounding hook: bar 2e-4 of the output's max (the bf16 kernel needs the hook and 2e-3)."""
    ws, H, W = 64, 128, 128
    hd = d // heads
    g = torch.Generator().manual_seed(5 * d + shift)
    qkv = torch.randn(H * W, 3 * d, generator=g)
    qkv[:, :2 * d] *= 1.3
    table = torch.randn(127 * 127, heads, generator=g) * 0.5
    out = ops.window_attention(qkv.cuda(), table.cuda(), 1, H, W, ws, shift, heads, precision="bf16x3").cpu()
    with torch.no_grad():
        t = qkv.view(1, H, W, 3 * d)
        if shift:
            t = torch.roll(t, (-shift, -shift), (1, 2))
        tw = R.window_partition(t, ws).view(-1, ws * ws, 3, heads, hd).permute(2, 0, 3, 1, 4)
        mask = R.calculate_mask(H, W, ws, shift) if shift else None
        outs = [R.attention_from_qkv(tw[0][i:i + 1] * hd ** -0.5, tw[1][i:i + 1], tw[2][i:i + 1], table, ws,
                                     None if mask is None else mask[i:i + 1]) for i in range(tw.shape[1])]
        o = torch.cat(outs).transpose(1, 2).reshape(-1, ws, ws, d)
        o = R.window_reverse(o, ws, H, W)
        if shift:
            o = torch.roll(o, (shift, shift), (1, 2))
        ref = o.reshape(H * W, d)
    e = float((out - ref).abs().max() / ref.abs().max())
    print(f"split-bf16 window-64 attention d={d} heads={heads} shift={shift}: rel err {e:.2e} vs the plain fp32 oracle")
    assert not torch.isnan(out).any() and e < 2e-4, e


def test_window64_attention_check_is_sensitive_to_a_bias_row_and_to_the_mask():
    """One ROW of the 127 x 127 relative-position table off (what a wrong ring slot would read), or the shift mask ignored, must
    fail the same comparison by >= 5x the bar."""
    d, heads, shift, ws, H, W = 212, 4, 32, 64, 128, 128
    sd = make(d, heads, seed=5, table_std=1.0)
    x = torch.randn(H * W, d, generator=torch.Generator().manual_seed(2)) * 1.5
    c = lambda k: sd[k].cuda()
    qkv_h = ops.ln_qkv(x.cuda(), c("norm1.weight"), c("norm1.bias"), c("qkv.weight"), c("qkv.bias"), heads, ws=ws, shift=shift)
    with torch.no_grad():
        ref = oracle_attention(qkv_h.cpu(), sd["relative_position_bias_table"], H, W, ws, heads, d, shift, LOG2E)
    run = lambda table, s: ops.window_attention_bf16_in(qkv_h, table.cuda(), 1, H, W, ws, s, d).cpu()
    rel = lambda a: float((a - ref).abs().max() / ref.abs().max())
    assert rel(run(sd["relative_position_bias_table"], shift)) < BAR
    bad = sd["relative_position_bias_table"].clone().view(127, 127, heads)
    bad[70] = bad[71].clone()                                        # table row dy = 70 - 63 = +7 reads its neighbour's values
    e_row = rel(run(bad.view(-1, heads), shift))
    e_mask = rel(run(sd["relative_position_bias_table"], 0))        # no shift: no mask, and the windows sit elsewhere
    print(f"sensitivity: one table row wrong {e_row:.2e}, shift / mask ignored {e_mask:.2e} (bar {BAR:.0e})")
    assert e_row > 5 * BAR and e_mask > 20 * BAR
