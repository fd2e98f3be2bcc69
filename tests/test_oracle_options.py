"""CPU: the oracle's optional rounding hooks restate WHERE a bf16 kernel rounds, never WHAT it computes - with an identity
``rnd`` every option of ``attention_core`` must reproduce the plain softmax attention (src/drct.py:271-299)."""
import torch

from oracle import sr_ref as R


def test_attention_core_hooks_are_arithmetic_no_ops_without_rounding():
    torch.manual_seed(0)
    d, heads, ws = 64, 2, 8
    sd = {"qkv.weight": torch.randn(3 * d, d) * 0.2, "qkv.bias": torch.randn(3 * d) * 0.1,
          "relative_position_bias_table": torch.randn(225, heads)}
    x = torch.randn(4, 64, d)
    mask = R.calculate_mask(16, 16, ws, 4)
    ident = lambda t: t
    plain = R.attention_core(sd, "", x, ws, heads, mask)
    for kw in ({"rnd": ident}, {"rnd": ident, "q_fold": 1.4426950408889634}, {"rnd": ident, "online_chunk": 16},
               {"rnd": ident, "q_fold": 1.4426950408889634, "online_chunk": 8}):
        got = R.attention_core(sd, "", x, ws, heads, mask, **kw)
        assert float((got - plain).abs().max()) < 2e-5, kw
    # and with a real rounding the streaming form differs from the one-pass form only at rounding level
    bf = lambda t: t.bfloat16().float()
    a = R.attention_core(sd, "", x, ws, heads, mask, rnd=bf)
    b = R.attention_core(sd, "", x, ws, heads, mask, rnd=bf, online_chunk=16)
    assert 0 < float((a - b).abs().max()) < 2e-2 * float(plain.abs().max())
