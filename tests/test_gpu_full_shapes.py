"""GPU: the BASELINE configurations at their FULL shapes (round 1 exercised C3 at 1 x 3 x 16 x 12 and C4 at 2 RDG only):
  C3  DRN-L x4, RGB, LR [8,3,64,64] -> HR [8,3,256,256]: bf16 mode against the fp32 mode (which the goldens hold to the
      reference), PSNR bar; batch-slot invariance (image i of the batch == the same image run alone);
  C4  DRCT-L x4 training, 12 RDG, 8 images of 32 x 32 per GPU, bf16: one graphed step bit-equal to the eager step, loss and
      every parameter; the fp32 / bf16 gradient agreement at full depth.
The CPU oracle needs minutes at these sizes, so the comparisons are mode-vs-mode and property based (size-
independent properties at full size, oracle parity at the sizes it finishes in seconds)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_c3_drn_full_shape_bf16_vs_fp32_mode_and_batch_slot_invariance():
    from srad_amd import spec as S
    from tests.helpers import DRN_GAIN
    from tests.test_gpu_drn import build
    cfg = S.DRNConfig.for_scale(4, 3)
    sd = S.synth_state(S.drn_spec(cfg), seed=31, gain=DRN_GAIN, cfg=cfg)
    x = torch.from_numpy(S.synth_image("c3", (8, 3, 64, 64), seed=4)).cuda()
    with torch.no_grad():
        o32 = build(cfg, sd, "fp32")(x)
        m16 = build(cfg, sd, "bf16")
        o16 = m16(x)
        assert [tuple(o.shape) for o in o16] == [(8, 3, 64, 64), (8, 3, 128, 128), (8, 3, 256, 256)]
        for a, b in zip(o16, o32):
            mse = float(((a - b).double() ** 2).mean())
            psnr = 10 * np.log10(255.0 ** 2 / mse)
            print("C3 output", tuple(a.shape), "bf16 vs fp32 mode: psnr(255)", round(psnr, 1), "max abs", float((a - b).abs().max()))
            assert psnr > 45.0
        # batch-slot invariance: no cross-image term (the channel-attention pooling is per image).  Not bit-exact: the pool's
        # partial sums come out of the conv's epilogue, one row per row tile, and the tile height follows the batch size, so
        # the summation order of the per-image means differs in the last bits (a cross-image term would be whole units of 255)
        alone = m16(x[5:6])
        for j in range(3):
            dif = (alone[j][0] - o16[j][5]).abs()
            print("C3 batch-slot invariance, output", j, "max abs", float(dif.max()), "mean abs", float(dif.mean()))
            assert float(dif.max()) < 0.1 and float(dif.mean()) < 0.005
        again = m16(x)
        assert all(torch.equal(again[j], o16[j]) for j in range(3))                    # bit-reproducible


@pytest.mark.parametrize("depth", [12])
def test_c4_full_depth_graphed_step_equals_eager(depth):
    from srad_amd import spec as S
    from srad_amd.train import FusedAdam, GraphedTrainStep, train_step
    from tests.test_gpu_train import build_train
    cfg = S.DRCTConfig(in_chans=1, img_size=32, window_size=8, upscale=4, n_rdg=depth)
    sd = S.synth_state(S.drct_spec(cfg), seed=44, gain=1.0, cfg=cfg)
    x = torch.from_numpy(S.synth_image("c4", (8, 1, 32, 32), seed=2)).cuda()
    hr = torch.from_numpy(S.synth_image("c4/hr", (8, 1, 128, 128), seed=3)).cuda()
    runs = {}
    for mode in ("eager", "graph"):
        m = build_train(cfg, sd, "bf16")                                  # DropPath off: the masks are the only random input
        opt = FusedAdam(m, lr=1e-4)
        step = GraphedTrainStep(m, opt, warmup=2) if mode == "graph" else (lambda a, b: train_step(m, a, b, opt))
        losses = [step(x, hr) for _ in range(4)]
        torch.cuda.synchronize()
        runs[mode] = (torch.stack([l.double() for l in losses]).cpu(), m.flat_params.clone())
        if mode == "graph":
            assert len(step._graphs) == 1
        del m, opt, step
        torch.cuda.empty_cache()
    assert torch.equal(runs["graph"][0], runs["eager"][0]), (runs["graph"][0], runs["eager"][0])
    assert torch.equal(runs["graph"][1], runs["eager"][1])
    assert float(runs["eager"][0][-1]) < float(runs["eager"][0][0])


def test_c4_full_depth_bf16_gradients_close_to_fp32_mode():
    import torch.nn.functional as F
    from srad_amd import spec as S
    from tests.test_gpu_train import build_train
    cfg = S.DRCTConfig(in_chans=1, img_size=32, window_size=8, upscale=4, n_rdg=12)
    sd = S.synth_state(S.drct_spec(cfg), seed=45, gain=1.0, cfg=cfg)
    x = torch.from_numpy(S.synth_image("c4g", (8, 1, 32, 32), seed=2)).cuda()
    hr = torch.from_numpy(S.synth_image("c4g/hr", (8, 1, 128, 128), seed=3)).cuda()
    gs = {}
    for prec in ("fp32", "bf16"):
        m = build_train(cfg, sd, prec)
        F.l1_loss(m(x), hr).backward()
        gs[prec] = m.flat_grads.double().clone()
        del m
        torch.cuda.empty_cache()
    cos = float((gs["fp32"] * gs["bf16"]).sum() / (gs["fp32"].norm() * gs["bf16"].norm()))
    ratio = float(gs["bf16"].norm() / gs["fp32"].norm())
    print(f"C4 full depth: cosine(fp32-mode grad, bf16-mode grad) = {cos:.6f}, norm ratio {ratio:.4f}")
    assert cos > 0.99 and 0.9 < ratio < 1.1
