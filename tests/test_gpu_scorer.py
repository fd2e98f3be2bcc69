"""GPU: scorer kernels (u8 conversion, SSIM window sweep via summed-area tables, MSE/PSNR, validation
metrics, AUC) against the fixtures the reference's own src/metrics.py produced and against the oracle.
Bars: u8 / quantize bit-exact; SSIM within 2e-6 absolute (fp32 SSIM arithmetic, fp64 box sums);
AUC exact to 1e-12."""
import numpy as np
import pytest
import torch

from oracle import scorer_ref as O

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("tag", ["gray", "rgb"])
def test_score_pairs_matches_reference_golden(scorer_golden, tag):
    from srad_amd import metrics as M
    g = scorer_golden
    sr, hr, wss = g[f"{tag}/sr"], g[f"{tag}/hr"], [int(w) for w in g[f"{tag}/ws"]]
    ssim, mse, psnr = M.score_pairs(torch.from_numpy(sr).cuda(), torch.from_numpy(hr).cuda(), wss)
    assert np.abs(ssim.cpu().numpy() - g[f"{tag}/ssim"]).max() < 2e-6
    assert np.abs(psnr.cpu().numpy() - g[f"{tag}/psnr"]).max() < 1e-5
    ref_mse = [float(np.mean((s.astype(np.float32) / 255 - h.astype(np.float32) / 255) ** 2)) for s, h in zip(sr, hr)]
    assert np.abs(mse.cpu().numpy() - np.array(ref_mse)).max() < 1e-9
    # reference-named wrappers
    assert abs(M.ssim_numpy(hr[0].astype(np.float32) / 255.0, sr[0].astype(np.float32) / 255.0, wss[1]) - g[f"{tag}/ssim"][0, 1]) < 2e-6
    assert abs(M.psnr_numpy(hr[1], sr[1]) - g[f"{tag}/psnr"][1]) < 1e-5


@pytest.mark.parametrize("tag", ["gray", "rgb"])
def test_validation_metrics_and_quantize(scorer_golden, tag):
    from srad_amd import metrics as M
    g = scorer_golden
    val_sr = torch.from_numpy(g[f"{tag}/val_sr"]).cuda()
    hr = torch.from_numpy(np.transpose(g[f"{tag}/hr"], (0, 3, 1, 2)).astype(np.float32)).cuda()
    psnr, ssim = M.val_metrics(val_sr, hr, 255.0)
    assert np.abs(psnr.cpu().numpy() - g[f"{tag}/val_psnr"]).max() < 1e-4
    assert np.abs(ssim.cpu().numpy() - g[f"{tag}/val_ssim"]).max() < 1e-6
    assert abs(M.psnr_torch(val_sr[:1], hr[:1], 255) - g[f"{tag}/val_psnr"][0]) < 1e-4
    q = M.quantize(val_sr * 1.003 - 0.2, 255.0).cpu().numpy()
    assert np.array_equal(q, g[f"{tag}/val_quant"])


def test_u8_truncation_and_rounding():
    from srad_amd import metrics as M
    x = torch.tensor([0.4, 0.5, 1.5, 2.5, 254.9, 255.4, 300.0, -3.0, 17.999], device="cuda").view(1, 1, 3, 3)
    assert M.to_u8_hwc(x).flatten().tolist() == O.to_u8_trunc(x.cpu().numpy()).flatten().tolist() == [0, 0, 1, 2, 254, 255, 255, 0, 17]
    assert M.quantize(x).flatten().tolist() == O.quantize_round(x.cpu().numpy()).flatten().tolist()
    rgb = torch.rand(2, 3, 5, 7, device="cuda") * 300 - 20
    assert np.array_equal(M.to_u8_hwc(rgb).cpu().numpy(), np.transpose(O.to_u8_trunc(rgb.cpu().numpy()), (0, 2, 3, 1)))
    assert np.array_equal(M.to_u8_hwc(rgb / 255, rgb_range=1.0).cpu().numpy(),
                          np.transpose(O.to_u8_trunc((rgb / 255).cpu().numpy(), 1.0), (0, 2, 3, 1)))


def test_auc_matches_sklearn_golden(scorer_golden):
    from srad_amd import metrics as M
    g = scorer_golden
    for c in sorted({k.split("/")[1] for k in g.files if k.startswith("auc/")}):
        assert abs(M.roc_auc(g[f"auc/{c}/y"], g[f"auc/{c}/s"]) - float(g[f"auc/{c}/auc"])) < 1e-12, c
    with pytest.raises(ValueError, match="one class"):
        M.roc_auc([1, 1, 1], [0.1, 0.2, 0.3])
    assert M.roc_auc([0, 1, 1], [0.5, float("inf"), 0.7]) == 1.0


@pytest.mark.parametrize("size,ch", [(128, 1), (64, 3), (33, 1)])
def test_full_sweep_vs_oracle(size, ch):
    """The evaluator's whole window sweep + three AUCs on synthetic good/bad pairs (MVTec-grid sized
    test split: 21 good + 57 bad at 128 px) against the CPU oracle."""
    from srad_amd import metrics as M
    n_good, n_bad = (21, 57) if size == 128 else (5, 7)
    y, sr, hr = O.synth_pairs(n_good, n_bad, size, ch, seed=0)
    ref = O.evaluate_pairs(y, sr, hr)
    got = M.evaluate_pairs(y, torch.from_numpy(np.stack(sr)).cuda(), torch.from_numpy(np.stack(hr)).cuda())
    assert got["window_sizes"] == ref["window_sizes"]
    assert got["best_ws"] == ref["best_ws"]
    assert np.abs(np.array(got["sweep_auc"]) - np.array(ref["sweep_auc"])).max() < 1e-12
    for k in ("auc_ssim", "auc_mse", "auc_psnr"):
        assert abs(got[k] - ref[k]) < 2e-3 and round(got[k], 3) == round(ref[k], 3), k
    assert np.abs(np.array(got["scores_ssim"]) - np.array(ref["scores_ssim"])).max() < 2e-6
    assert np.abs(np.array(got["scores_mse"]) - np.array(ref["scores_mse"])).max() < 1e-9


def test_large_tile_properties():
    """1024 px tiles (config C5 scorer shape): properties that need no CPU reference - identical images
    score SSIM 1 / MSE 0 / PSNR inf for every window, and scores are invariant to which slot of the
    batch a pair sits in (the chunked summed-area tables do not leak between images)."""
    from srad_amd import metrics as M
    g = torch.Generator(device="cpu").manual_seed(0)
    hr = (torch.rand(3, 1024, 1024, 1, generator=g) * 255).to(torch.uint8).cuda()
    sr = hr.clone()
    sr[1] = (sr[1].float() * 0.9).to(torch.uint8)
    sizes = [3, 503, 1021]
    ssim, mse, psnr = M.score_pairs(sr, hr, sizes)
    assert torch.all((ssim[0] - 1).abs() < 1e-12) and mse[0] == 0 and torch.isinf(psnr[0])
    assert torch.all(ssim[1] < 0.999) and mse[1] > 0
    ssim2, mse2, _ = M.score_pairs(sr[[1, 0, 2]].contiguous(), hr[[1, 0, 2]].contiguous(), sizes)
    assert torch.equal(ssim2[0], ssim[1]) and torch.equal(ssim2[1], ssim[0]) and torch.equal(mse2[0], mse[1])


def test_large_tile_full_sweep_is_symmetric_and_batch_invariant():
    """The C5 scorer shape in full - 1024 px tiles, all 102 window sizes of the reference's sweep (src/evaluate.py:149-150) - through
    properties that need no CPU reference: MSE and PSNR are symmetric in their two images bit for bit, SSIM to the last fp32 bits
    of its per-pixel map (mu1^2 + mu2^2 is contracted into one fma, which rounds the two squares differently: 1e-7, the parity bar
    is 2e-6), and a pair scores the same alone as inside a batch."""
    from srad_amd import metrics as M
    g = torch.Generator(device="cpu").manual_seed(3)
    hr = (torch.rand(2, 1024, 1024, 1, generator=g) * 255).to(torch.uint8).cuda()
    sr = (hr.int() + torch.randint(-9, 10, hr.shape, generator=g).cuda()).clamp(0, 255).to(torch.uint8)
    sizes = M.sweep_window_sizes(1024)
    assert len(sizes) == 102
    a = M.score_pairs(sr, hr, sizes)
    b = M.score_pairs(hr, sr, sizes)
    assert float((a[0] - b[0]).abs().max()) < 1e-7 and torch.equal(a[1], b[1]) and torch.equal(a[2], b[2])
    one = M.score_pairs(sr[1:2].contiguous(), hr[1:2].contiguous(), sizes)
    assert torch.equal(one[0][0], a[0][1]) and torch.equal(one[1][0], a[1][1])
    assert bool(((a[0] > 0) & (a[0] < 1)).all()) and bool((a[0][:, 1:] != a[0][:, :-1]).any())


def test_l1_loss():
    from srad_amd import metrics as M
    a, b = torch.randn(3, 1, 37, 41, device="cuda"), torch.randn(3, 1, 37, 41, device="cuda")
    assert abs(M.l1_loss(a, b).item() - (a - b).abs().double().mean().item()) < 1e-9
