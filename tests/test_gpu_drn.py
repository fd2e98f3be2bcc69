"""GPU: DRN-L forward + dual regression model through the C ABI against the fixtures the reference
produced.  Bars as for DRCT: fp32 mode <= 1e-3 relative (measured ~1e-5); bf16 mode PSNR >= 35 dB."""
import numpy as np
import pytest
import torch

from tests.helpers import DRN_CASES, drn_case, rel_err

pytestmark = pytest.mark.gpu


class Opt:
    def __init__(self, cfg, precision, use_graph=False):
        self.n_colors, self.n_blocks, self.n_feats, self.negval, self.rgb_range = cfg.n_colors, cfg.n_blocks, cfg.n_feats, cfg.negval, cfg.rgb_range
        self.scale = cfg.scales
        self.precision, self.use_graph = precision, use_graph


def build(cfg, sd, precision, use_graph=False):
    from srad_amd.nets import DRN
    m = DRN(Opt(cfg, precision, use_graph)).cuda().eval()
    m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}, strict=True)
    return m


@pytest.mark.parametrize("name", DRN_CASES)   # incl. the x8 preset: 10 features, stored as 12 at level 0
@pytest.mark.parametrize("graph", [False, True])
def test_drn_fp32_matches_reference_golden(sr_golden, name, graph):
    cfg, sd, dual, x, ys, dual_y = drn_case(sr_golden, name)
    m = build(cfg, sd, "fp32", use_graph=graph)
    with torch.no_grad():
        for _ in range(3 if graph else 1):
            outs = m(torch.from_numpy(x).cuda())
    assert len(outs) == cfg.phase + 1
    for o, y in zip(outs, ys):
        assert tuple(o.shape) == y.shape
        assert rel_err(o.cpu().numpy(), y) < 2e-4
    # dual regression model on the finest output
    from srad_amd.nets import DownBlock
    d = DownBlock(Opt(cfg, "fp32")).cuda()
    d.load_state_dict({k: torch.from_numpy(v) for k, v in dual.items()})
    with torch.no_grad():
        dy = d(torch.from_numpy(ys[-1]).cuda())
    assert rel_err(dy.cpu().numpy(), dual_y) < 2e-4


@pytest.mark.parametrize("name", DRN_CASES)
def test_drn_split_bf16_matches_reference_golden(sr_golden, name):
    """split-bf16 ("bf16x3"): every convolution through the split GEMM (hi + lo planes of A and W), held to the fp32 bar;
    the dual model too."""
    cfg, sd, dual, x, ys, dual_y = drn_case(sr_golden, name)
    m = build(cfg, sd, "bf16x3")
    with torch.no_grad():
        outs = m(torch.from_numpy(x).cuda())
    for o, y in zip(outs, ys):
        assert tuple(o.shape) == y.shape
        assert rel_err(o.cpu().numpy(), y) < 2e-4, rel_err(o.cpu().numpy(), y)
    from srad_amd.nets import DownBlock
    d = DownBlock(Opt(cfg, "bf16x3")).cuda()
    d.load_state_dict({k: torch.from_numpy(v) for k, v in dual.items()})
    with torch.no_grad():
        dy = d(torch.from_numpy(ys[-1]).cuda())
    assert rel_err(dy.cpu().numpy(), dual_y) < 2e-4


@pytest.mark.parametrize("name", ["drn_x2_gray", "drn_x4_rgb"])
def test_drn_bf16_close_to_reference(sr_golden, name):
    cfg, sd, dual, x, ys, _ = drn_case(sr_golden, name)
    m = build(cfg, sd, "bf16")
    with torch.no_grad():
        outs = m(torch.from_numpy(x).cuda())
    y = ys[-1]
    out = outs[-1].cpu().numpy()
    psnr = 10 * np.log10(255.0 ** 2 / np.mean((out - y).astype(np.float64) ** 2))
    print(name, "bf16 max abs err", np.abs(out - y).max(), "psnr(255)", psnr)
    assert psnr > 35.0


def test_drn_x8_preset_bf16_and_odd_feats_rejected(sr_golden):
    """x8 preset (n_feats=10 -> level 0 zero-padded to 12) in bf16; an odd n_feats is refused with a message."""
    cfg, sd, dual, x, ys, _ = drn_case(sr_golden, "drn_x8_gray")
    with torch.no_grad():
        out = build(cfg, sd, "bf16")(torch.from_numpy(x).cuda())[-1].cpu().numpy()
    psnr = 10 * np.log10(255.0 ** 2 / np.mean((out - ys[-1]).astype(np.float64) ** 2))
    assert psnr > 35.0
    from srad_amd import spec as S
    from srad_amd.nets import DRN
    import dataclasses
    bad = dataclasses.replace(S.DRNConfig.for_scale(4, 1), n_feats=7)
    with pytest.raises(RuntimeError, match="n_feats must be even"):
        DRN(Opt(bad, "fp32")).cuda().eval()(torch.zeros(1, 1, 4, 4, device="cuda"))


def test_drn_oracle_c1_shape():
    """BASELINE config C1 shape: DRN-L x2, gray, batch 2, 32x32 LR -> [[2,1,32,32],[2,1,64,64]]."""
    from oracle import sr_ref as R
    from srad_amd import spec as S
    cfg = S.DRNConfig.for_scale(2, 1)
    sd = S.synth_state(S.drn_spec(cfg), seed=5, gain=0.5, cfg=cfg)
    x = S.synth_image("c1", (2, 1, 32, 32), seed=1)
    with torch.no_grad():
        ref = R.drn_forward(sd, torch.from_numpy(x), cfg)
        out = build(cfg, sd, "fp32")(torch.from_numpy(x).cuda())
    assert [tuple(o.shape) for o in out] == [(2, 1, 32, 32), (2, 1, 64, 64)]
    for o, r in zip(out, ref):
        assert rel_err(o.cpu().numpy(), r.numpy()) < 2e-4
