"""CPU: pin the oracle (oracle/sr_ref.py, oracle/scorer_ref.py) against fixtures produced by the
reference itself (tests/golden/make_golden.py)."""
import numpy as np
import pytest
import torch

from oracle import scorer_ref as O
from oracle import sr_ref as R
from srad_amd import spec as S
from tests.helpers import DRCT_CASES, DRN_CASES, drct_case, drn_case, rel_err


@pytest.mark.parametrize("name", DRCT_CASES)
def test_drct_forward_matches_reference(sr_golden, name):
    cfg, sd, x, y = drct_case(sr_golden, name)
    taps = {}
    with torch.no_grad():
        out = R.drct_forward(sd, torch.from_numpy(x), cfg, taps=taps).numpy()
    assert out.shape == y.shape
    assert rel_err(out, y) < 2e-5
    for k in ("embed", "rdg0"):
        key = f"{name}/tap_{k}"
        if key in sr_golden:
            step = int(sr_golden[key + "_step"])
            assert rel_err(taps[k][:, ::step].numpy(), sr_golden[key]) < 2e-5


def test_drct_grads_match_reference(sr_golden):
    name = "drct_r2_rgb_x4"
    cfg, sd, x, _ = drct_case(sr_golden, name)
    params = {k: torch.from_numpy(v).clone().requires_grad_(v.dtype == np.float32 and "mask" not in k)
              for k, v in sd.items()}
    xg = torch.from_numpy(x).clone().requires_grad_(True)
    loss = R.l1(R.drct_forward(params, xg, cfg), torch.from_numpy(sr_golden[name + "/hr"]))
    loss.backward()
    assert abs(loss.item() - float(sr_golden[name + "/loss"])) < 1e-4 * abs(loss.item())
    assert rel_err(xg.grad.numpy(), sr_golden[name + "/grad_x"]) < 1e-4
    for key in sr_golden.files:
        if key.startswith(name + "/grad/"):
            k = key[len(name + "/grad/"):]
            assert rel_err(params[k].grad.numpy(), sr_golden[key]) < 1e-4, k
    names = list(sr_golden[name + "/grad_names"])
    l2 = sr_golden[name + "/grad_l2"]
    for n, ref in zip(names, l2):
        got = float(params[str(n)].grad.double().pow(2).sum().sqrt())
        assert abs(got - ref) <= 1e-4 * max(ref, 1e-8), n


@pytest.mark.parametrize("name", DRN_CASES)
def test_drn_forward_matches_reference(sr_golden, name):
    cfg, sd, dual, x, ys, dual_y = drn_case(sr_golden, name)
    with torch.no_grad():
        outs = R.drn_forward(sd, torch.from_numpy(x), cfg)
        d = R.dual_forward(dual, outs[-1], cfg)
    assert len(outs) == cfg.phase + 1
    for o, y in zip(outs, ys):
        assert o.shape == y.shape
        assert rel_err(o.numpy(), y) < 2e-5
    assert rel_err(d.numpy(), dual_y) < 2e-5


def test_spec_sizes_match_survey():
    # DRCT x4 gray = 1000 entries / 27.38 M params, DRN x4 = 664, x2 = 368
    sp = S.drct_spec(S.DRCTConfig())
    assert len(sp) == 1000
    n = sum(int(np.prod(s)) for s, k in sp.values() if k not in ("index", "mask"))
    assert n == 27382021
    assert len(S.drn_spec(S.DRNConfig.for_scale(4, 3))) == 664
    assert len(S.drn_spec(S.DRNConfig.for_scale(2, 1))) == 368
    assert S.DRCTConfig().block_table() == [(180, 6, 360, 0), (212, 4, 424, 4), (244, 2, 488, 0),
                                           (276, 6, 276, 4), (308, 4, 308, 0)]


def test_mask_and_index_buffers():
    for ws in (2, 4, 8, 16):
        assert np.array_equal(S.relative_position_index(ws), R.rel_pos_index(ws).numpy())
        H = W = 4 * ws
        assert np.array_equal(S.shifted_window_mask(H, W, ws, ws // 2), R.calculate_mask(H, W, ws, ws // 2).numpy())


@pytest.mark.parametrize("tag", ["gray", "rgb"])
def test_scorer_matches_reference(scorer_golden, tag):
    g = scorer_golden
    sr, hr, wss = g[f"{tag}/sr"], g[f"{tag}/hr"], g[f"{tag}/ws"]
    for i in range(len(sr)):
        sf, hf = sr[i].astype(np.float32) / 255.0, hr[i].astype(np.float32) / 255.0
        for j, ws in enumerate(wss):
            ref = g[f"{tag}/ssim"][i, j]
            assert abs(O.ssim_numpy(hf, sf, int(ws)) - ref) < 2e-6
            if i == 0 and ws <= 13:
                assert abs(O.ssim_numpy(hf, sf, int(ws), fast=False) - ref) < 1e-7
        assert abs(O.psnr_numpy(hf, sf) - g[f"{tag}/psnr"][i]) < 1e-5
        # integer input (not used by evaluate.py): the reference picks data_range AFTER its float32
        # cast, so C1/C2 stay at the [0,1] values; float32 cancellation in E[x^2]-mu^2 at 0..255
        # makes the result summation-order sensitive, so only the literal loop is held tight
        assert abs(O.ssim_numpy(hr[i], sr[i], 7, fast=False) - g[f"{tag}/ssim_u8"][i]) < 1e-6
        assert abs(O.ssim_numpy(hr[i], sr[i], 7) - g[f"{tag}/ssim_u8"][i]) < 2e-3
    val_sr = g[f"{tag}/val_sr"]
    ht = np.transpose(hr, (0, 3, 1, 2)).astype(np.float32)
    for i in range(len(sr)):
        assert abs(O.psnr_torch_ref(val_sr[i:i + 1], ht[i:i + 1], 255) - g[f"{tag}/val_psnr"][i]) < 1e-4
        assert abs(O.ssim_torch_ref(val_sr[i:i + 1], ht[i:i + 1], 255) - g[f"{tag}/val_ssim"][i]) < 1e-6
    assert np.array_equal(O.quantize_round(val_sr * np.float32(1.003) - np.float32(0.2), 255), g[f"{tag}/val_quant"])


def test_auc_matches_sklearn(scorer_golden):
    g = scorer_golden
    cases = sorted({k.split("/")[1] for k in g.files if k.startswith("auc/")})
    assert len(cases) >= 6
    for c in cases:
        assert abs(O.roc_auc(g[f"auc/{c}/y"], g[f"auc/{c}/s"]) - float(g[f"auc/{c}/auc"])) < 1e-12, c
    with pytest.raises(ValueError):
        O.roc_auc([1, 1, 1], [0.1, 0.2, 0.3])


def test_sweep_sizes():
    # 13 sizes @128, 6 @64, 26 @256, 102 @1024
    assert [len(O.sweep_window_sizes(n)) for n in (128, 64, 256, 1024)] == [13, 6, 26, 102]
    assert O.sweep_window_sizes(128)[:3] == [3, 13, 23]
    assert O.sweep_window_sizes(4) == [3]


def test_truncate_vs_round():
    x = np.array([0.4, 0.5, 1.5, 2.5, 254.9, 255.4, 300.0, -3.0], dtype=np.float32)
    assert O.to_u8_trunc(x).tolist() == [0, 0, 1, 2, 254, 255, 255, 0]
    assert O.quantize_round(x).tolist() == [0.0, 0.0, 2.0, 2.0, 255.0, 255.0, 255.0, 0.0]
