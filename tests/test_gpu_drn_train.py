"""GPU: DRN-L training step through the C ABI (DRN branch; B1-B6 backward): gradients of every
parameter (incl. the trainable MeanShift layers, H4) and of the dual regression models under the reference's
composite loss against torch autograd of the oracle; fused Adam steps reduce the loss.
fp32 mode, bar 1e-3 relative per tensor (max error over the tensor's max)."""
import numpy as np
import pytest
import torch

from tests.helpers import DRN_GAIN, rel_err
from tests.test_gpu_drn import Opt

pytestmark = pytest.mark.gpu


def _setup(scale, n_colors, n_blocks, n_feats, B, H, W, seed=31, prec="fp32"):
    from srad_amd import spec as S
    from srad_amd.nets import DRN, DownBlock
    cfg = S.DRNConfig(n_colors=n_colors, scale=scale, n_blocks=n_blocks, n_feats=n_feats)
    sd = S.synth_state(S.drn_spec(cfg), seed=seed, gain=DRN_GAIN, cfg=cfg)
    duals = [S.synth_state(S.dual_spec(cfg), seed=seed + 100 + i, gain=DRN_GAIN, cfg=cfg) for i in range(cfg.phase)]
    x = S.synth_image("drn_tr", (B, n_colors, H, W), seed=5)
    lrs = [x] + [S.synth_image(f"drn_tr/lr{i}", (B, n_colors, H * 2 ** i, W * 2 ** i), seed=6 + i) for i in range(1, cfg.phase)]
    hr = S.synth_image("drn_tr/hr", (B, n_colors, H * scale, W * scale), seed=9)
    m = DRN(Opt(cfg, prec)).cuda()
    m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}, strict=True)
    m.train()
    m.enable_training()
    dms = []
    for d in duals:
        dm = DownBlock(Opt(cfg, prec)).cuda()
        dm.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in d.items()})
        dms.append(dm)
    return cfg, sd, duals, lrs, hr, m, dms


@pytest.mark.parametrize("scale,n_colors,n_feats,B,H,W", [(4, 1, 20, 2, 8, 12), (2, 3, 8, 1, 16, 16), (4, 3, 20, 1, 8, 8),
                                                             (8, 1, 10, 2, 4, 6), (8, 3, 10, 1, 4, 4), (2, 1, 10, 1, 8, 8)])
def test_drn_gradients_match_oracle_autograd(scale, n_colors, n_feats, B, H, W):
    from oracle import sr_ref as R
    from srad_amd.train import drn_loss
    cfg, sd, duals, lrs, hr, m, dms = _setup(scale, n_colors, 2, n_feats, B, H, W)
    # oracle: torch CPU autograd over the same state
    sdt = {k: torch.from_numpy(np.asarray(v)).clone().requires_grad_(True) for k, v in sd.items()}
    dts = [{k: torch.from_numpy(np.asarray(v)).clone().requires_grad_(True) for k, v in d.items()} for d in duals]
    sr_ref = R.drn_forward(sdt, torch.from_numpy(lrs[0]), cfg)
    sr2lr_ref = [R.dual_forward(dts[i], sr_ref[i - len(dts)], cfg) for i in range(len(dts))]
    loss_ref = R.drn_total_loss(sr_ref, [torch.from_numpy(a) for a in lrs], torch.from_numpy(hr), sr2lr_ref)
    loss_ref.backward()
    # engine
    lr_t = [torch.from_numpy(a).cuda() for a in lrs]
    sr = m(lr_t[0])
    assert len(sr) == cfg.phase + 1
    for a, b in zip(sr, sr_ref):
        assert rel_err(a.detach().cpu().numpy(), b.detach().numpy()) < 2e-4
    sr2lr = [dms[i](sr[i - len(dms)]) for i in range(len(dms))]
    loss = drn_loss(sr, lr_t, torch.from_numpy(hr).cuda(), sr2lr)
    assert abs(float(loss.detach()) - float(loss_ref.detach())) < 1e-5 * abs(float(loss_ref.detach()))
    loss.backward()
    worst = ("", 0.0)
    for n, p in m.named_parameters():
        e = rel_err(p.grad.cpu().numpy(), sdt[n].grad.numpy())
        worst = max(worst, (n, e), key=lambda t: t[1])
        assert e < 1e-3, (n, e)
    for dm, dt in zip(dms, dts):
        for n, p in dm.named_parameters():
            e = rel_err(p.grad.cpu().numpy(), dt[n].grad.numpy())
            assert e < 1e-3, ("dual " + n, e)
    print("worst DRN parameter-gradient error:", worst)


def test_drn_train_steps_reduce_the_loss_and_eval_uses_new_weights():
    from srad_amd.train import FusedAdam, drn_train_step
    cfg, sd, duals, lrs, hr, m, dms = _setup(4, 1, 2, 20, 2, 8, 8)
    opt = FusedAdam(m, lr=1e-4, weight_decay=1e-8)                      # src/main.py:61-66 (DRN: weight decay 1e-8)
    dopts = [torch.optim.Adam(dm.parameters(), lr=1e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-8) for dm in dms]
    lr_t = [torch.from_numpy(a).cuda() for a in lrs]
    hr_t = torch.from_numpy(hr).cuda()
    before = m.flat_params.clone()
    losses = [float(drn_train_step(m, dms, lr_t, hr_t, opt, dopts)) for _ in range(5)]
    print("DRN losses", losses)
    assert losses[-1] < losses[0]
    assert float((m.flat_params - before).abs().max()) > 0
    m.eval()
    with torch.no_grad():
        y = m(lr_t[0])
    assert len(y) == 3 and bool(torch.isfinite(y[-1]).all())


@pytest.mark.parametrize("prec", ["fp32", "bf16"])
def test_graphed_drn_step_equals_the_eager_steps(prec):
    """GraphedDrnTrainStep (one hipGraph per step: forward, composite loss, autograd through the dual models, two-stream
    backward, Adam for the SR net and the dual models) against the same steps launched eagerly: same losses, same weights."""
    from srad_amd.train import FusedAdam, GraphedDrnTrainStep, TensorAdam, drn_train_step
    runs = []
    for mode in ("eager", "graph"):
        cfg, sd, duals, lrs, hr, m, dms = _setup(4, 3, 2, 20, 2, 16, 16, prec=prec)
        opt = FusedAdam(m, lr=1e-4, weight_decay=1e-8)
        dopts = [TensorAdam(dm.parameters(), lr=1e-4, weight_decay=1e-8) for dm in dms]
        lr_t = [torch.from_numpy(a).cuda() for a in lrs]
        hr_t = torch.from_numpy(hr).cuda()
        step = (GraphedDrnTrainStep(m, dms, opt, dopts, warmup=2) if mode == "graph"
                else (lambda a, b: drn_train_step(m, dms, a, b, opt, dopts)))
        losses = [float(step([t + 0.5 * i for t in lr_t], hr_t)) for i in range(6)]       # a different batch every step
        if mode == "graph":                 # what the reference's Loss log adds up per step: every term, the dual ones unweighted
            assert step.logged is not None and float(step.logged) > losses[-1] > 0
        runs.append((losses, m.flat_params.clone(), [p.detach().clone() for dm in dms for p in dm.parameters()],
                     opt.step_count, [o.step_count for o in dopts]))
    (le, pe, de, se, sde), (lg, pg, dg, sg, sdg) = runs
    print("eager", le, "graph", lg)
    assert se == sg == 6 and sde == sdg
    assert max(abs(a - b) / abs(a) for a, b in zip(le, lg)) < 1e-5
    assert rel_err(pg.cpu().numpy(), pe.cpu().numpy()) < 1e-5
    for a, b in zip(de, dg):
        assert rel_err(b.cpu().numpy(), a.cpu().numpy()) < 1e-5


@pytest.mark.parametrize("prec", ["fp32", "bf16"])
def test_drn_x8_preset_trains(prec):
    """The x8 preset (src/main.py:297-300: n_feats 10, three phases) stores level 0 zero-padded to 12 channels; the flat
    parameter / gradient buffers keep the reference's tensors.  Steps reduce the loss, the pad columns stay out of the
    state dict, and the inference engine sees the trained weights."""
    from srad_amd.train import FusedAdam, TensorAdam, drn_train_step
    cfg, sd, duals, lrs, hr, m, dms = _setup(8, 1, 2, 10, 2, 4, 4, prec=prec)
    assert {k: tuple(v.shape) for k, v in m.state_dict().items()} == {k: tuple(np.asarray(v).shape) for k, v in sd.items()}
    opt = FusedAdam(m, lr=1e-4, weight_decay=1e-8)
    dopts = [TensorAdam(dm.parameters(), lr=1e-4, weight_decay=1e-8) for dm in dms]
    lr_t = [torch.from_numpy(a).cuda() for a in lrs]
    hr_t = torch.from_numpy(hr).cuda()
    losses = [float(drn_train_step(m, dms, lr_t, hr_t, opt, dopts)) for _ in range(5)]
    print("DRN x8 losses", prec, losses)
    assert losses[-1] < losses[0]
    m.eval()
    with torch.no_grad():
        y = m(lr_t[0])
    assert len(y) == 4 and y[-1].shape[-1] == 32 and bool(torch.isfinite(y[-1]).all())
    if prec == "fp32":                       # eval forward of the trained weights == oracle forward of the same state
        from oracle import sr_ref as R
        st = {k: v.detach().cpu() for k, v in m.state_dict().items()}
        ref = R.drn_forward(st, torch.from_numpy(lrs[0]), cfg)
        assert rel_err(y[-1].cpu().numpy(), ref[-1].numpy()) < 2e-4


@pytest.mark.parametrize("scale,n_feats", [(4, 20), (8, 10)])
def test_drn_gradient_buckets_are_final_when_the_hook_fires(scale, n_feats):
    """srad_drn_backward's bucket hook (data-parallel all-reduce overlapped with the backward, as DRCT's): the buckets tile the
    flat gradient buffer, fire in order 0 .. phase + 1, and a stream-ordered copy taken inside the hook already holds the
    bucket's final values."""
    from srad_amd.train import drn_loss
    cfg, sd, duals, lrs, hr, m, dms = _setup(scale, 1, 2, n_feats, 2, 4, 4)
    tot = m.flat_grads.numel()
    assert len(m.grad_buckets) == cfg.phase + 2
    spans = sorted(m.grad_buckets)
    assert spans[0][0] == 0 and all(a + n == b for (a, n), (b, _) in zip(spans, spans[1:])) and sum(spans[-1]) == tot
    fired, snaps = [], []

    def hook(b):
        off, n = m.grad_buckets[b]
        fired.append(b)
        snaps.append(m.flat_grads[off:off + n].clone())
    m.on_bucket = hook
    lr_t = [torch.from_numpy(a).cuda() for a in lrs]
    sr = m(lr_t[0])
    sr2lr = [dms[i](sr[i - len(dms)]) for i in range(len(dms))]
    drn_loss(sr, lr_t, torch.from_numpy(hr).cuda(), sr2lr).backward()
    torch.cuda.synchronize()
    assert fired == list(range(cfg.phase + 2))
    for b, snap in zip(fired, snaps):
        off, n = m.grad_buckets[b]
        assert float(snap.abs().max()) > 0 and torch.equal(snap, m.flat_grads[off:off + n]), b


def test_cli_train_drn_writes_run_dir_with_dual_models(tmp_path):
    """src/main.py train_drn on the folder layout src/data.py reads (HR + LR_2 + LR_4): dual models, their Adam optimizers
    (the engine's kernel, tensor by tensor) and cosine schedules, run-dir files incl. dual_model_*.pt and dual_optimizers.pt;
    the composite loss falls over two virtual epochs."""
    import os
    from PIL import Image
    from srad_amd import main as Mn
    from srad_amd import options as Opt
    rng = np.random.default_rng(0)
    yy, xx = np.mgrid[0:64, 0:64]
    for split, n in (("train", 4), ("val", 1)):
        d = tmp_path / "data" / "grid" / split / "good"
        for sub in ("HR", "LR_2", "LR_4"):
            (d / sub).mkdir(parents=True)
        for i in range(n):
            hr = (127 + 90 * np.sin(xx / (3.0 + i)) * np.cos(yy / 4.0) + rng.normal(0, 4, (64, 64))).clip(0, 255).astype(np.uint8)
            Image.fromarray(hr).save(d / "HR" / f"{i}.png")
            Image.fromarray(hr.reshape(32, 2, 32, 2).mean((1, 3)).round().astype(np.uint8)).save(d / "LR_2" / f"{i}.png")
            Image.fromarray(hr.reshape(16, 4, 16, 4).mean((1, 3)).round().astype(np.uint8)).save(d / "LR_4" / f"{i}.png")
    args = Opt.parse_train_args(["--model-type", "drn-l", "--classe", "grid", "--resolution", "64", "--scale", "4", "--epochs", "2",
                                 "--batch-size", "2", "--data-root", str(tmp_path / "data"), "--save-dir", str(tmp_path / "exp")])
    opt = Mn.build_train_opt(args)
    assert opt.scale == [2, 4] and opt.weight_decay == 1e-8 and (opt.n_blocks, opt.n_feats) == (40, 20) and opt.test_every == 128
    opt.n_blocks, opt.test_every, opt.print_every = 2, 4, 2
    Mn.train_drn(opt)
    run = opt.save
    assert {"config.txt", "log.txt", "loss_log.pt", "optimizer.pt", "dual_optimizers.pt", "psnr_ssim_log.pt"} <= set(os.listdir(run))
    assert sorted(os.listdir(os.path.join(run, "model"))) == ["dual_model_best.pt", "dual_model_latest.pt", "model_best.pt", "model_latest.pt"]
    duals = torch.load(os.path.join(run, "model", "dual_model_latest.pt"), weights_only=True)
    assert isinstance(duals, list) and len(duals) == 2 and set(duals[0]) == {"dual_module.0.0.weight", "dual_module.1.weight"}
    dopt = torch.load(os.path.join(run, "dual_optimizers.pt"))
    assert sorted(dopt) == [0, 1] and dopt[0]["step"] == 8
    loss_log = torch.load(os.path.join(run, "loss_log.pt"))
    assert tuple(loss_log.shape) == (2, 1) and float(loss_log[1, 0]) < float(loss_log[0, 0])
    assert "scale: [2, 4]" in open(os.path.join(run, "config.txt")).read()


@pytest.mark.parametrize("lr_px", [16, 64])
def test_drn_bf16_gradients_close_to_fp32_mode(lr_px):
    """bf16 mode of the DRN-L training step (incl. the 80 x 80-tile weight-gradient kernel of the 80-channel RCAB convolutions,
    which only exists in bf16) against the fp32 mode on the same weights and batch: cosine of the flat gradient and the
    per-tensor relative L2 error of the RCAB convolution weights.  At 64 px LR the 80-channel convolutions - forward, and the
    data gradients with their ReLU' / skip-path epilogues - take the weight-resident kernel (kernels_conv80.hip), which the
    fp32 mode never does."""
    from srad_amd.nets import DRN
    from srad_amd.train import drn_loss
    cfg, sd, duals, lrs, hr, m32, dms = _setup(4, 3, 2, 20, 2, lr_px, lr_px)
    m16 = DRN(Opt(cfg, "bf16")).cuda()
    m16.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}, strict=True)
    m16.train()
    m16.enable_training()
    lr_t = [torch.from_numpy(a).cuda() for a in lrs]
    hr_t = torch.from_numpy(hr).cuda()
    grads = {}
    for name, m in (("fp32", m32), ("bf16", m16)):
        for dm in dms:
            dm.zero_grad()
        sr = m(lr_t[0])
        sr2lr = [dms[i](sr[i - len(dms)]) for i in range(len(dms))]
        drn_loss(sr, lr_t, hr_t, sr2lr).backward()
        grads[name] = (m.flat_grads.double().clone(), {n: p.grad.double().clone() for n, p in m.named_parameters()})
    a, b = grads["bf16"][0], grads["fp32"][0]
    cos = float((a * b).sum() / (a.norm() * b.norm()))
    worst = max(float((grads["bf16"][1][n] - g).norm() / g.norm().clamp_min(1e-30))
                for n, g in grads["fp32"][1].items() if ".body." in n and n.endswith("weight") and g.dim() == 4 and g.shape[0] == 80 == g.shape[1])
    print(f"DRN bf16 vs fp32 mode: gradient cosine {cos:.6f}, worst relative L2 over the 80 -> 80 conv weights {worst:.3e}")
    assert cos > 0.9999 and worst < 0.1          # bf16 forward + backward chain; the kernel itself is held to 2e-2 in test_gpu_bwd_ops.py
