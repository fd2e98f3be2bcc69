"""GPU: the DRCT training step through the C ABI.
 * gradients under nn.L1Loss against the fixtures the reference's autograd produced (G7) - every parameter's
   L2 norm plus eight full tensors and dLoss/dx;
 * training-mode forward/backward with explicit DropPath masks against autograd of the oracle;
 * fused Adam step == torch.optim.Adam on the same gradients; loss goes down over a few steps;
 * bf16 mode gradients close to the fp32 ones.
Bars: fp32 mode 1e-3 relative (north_star), per-tensor max error relative to that tensor's max."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from tests.helpers import drct_case, rel_err
from tests.test_gpu_drct import Opt

pytestmark = pytest.mark.gpu


def build_train(cfg, sd, precision, drop_path_rate=0.0):
    from srad_amd.nets import DRCT
    o = Opt(cfg, precision, use_graph=False)
    o.drop_path_rate = drop_path_rate
    m = DRCT(o).cuda()
    m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}, strict=True)
    m.train()
    m.enable_training()
    return m


def test_gradients_match_reference_autograd_golden(sr_golden):
    g = sr_golden
    name = "drct_r2_rgb_x4"
    cfg, sd, x, y = drct_case(g, name)
    m = build_train(cfg, sd, "fp32")
    xt = torch.from_numpy(x).cuda().requires_grad_(True)
    hr = torch.from_numpy(g[name + "/hr"]).cuda()
    out = m(xt)
    assert rel_err(out.detach().cpu().numpy(), y) < 2e-4              # training forward == eval forward (no DropPath)
    loss = F.l1_loss(out, hr)
    assert abs(float(loss.detach()) - float(g[name + "/loss"])) < 1e-5 * abs(float(g[name + "/loss"]))
    loss.backward()
    assert rel_err(xt.grad.cpu().numpy(), g[name + "/grad_x"]) < 1e-3
    grads = dict(m.named_parameters())
    for k in [k for k in g.files if k.startswith(name + "/grad/")]:
        pname = k[len(name + "/grad/"):]
        e = rel_err(grads[pname].grad.cpu().numpy(), g[k])
        assert e < 1e-3, (pname, e)
    names = [str(n) for n in g[name + "/grad_names"]]
    l2 = g[name + "/grad_l2"]
    worst = 0.0
    for n, ref in zip(names, l2):
        mine = float(grads[n].grad.double().pow(2).sum().sqrt())
        worst = max(worst, abs(mine - ref) / max(ref, 1e-12))
        assert abs(mine - ref) <= 1e-3 * max(ref, 1e-9), (n, mine, ref)
    print("worst relative L2-norm error over", len(names), "parameter tensors:", worst)


@pytest.mark.parametrize("name", ["drct_r1_gray_x4_ws4", "drct_r1_gray_x8_ws2", "drct_r1_gray_x4_ws16"])
def test_gradients_match_reference_autograd_golden_other_window_sizes(sr_golden, name):
    """G7 extended (round 3, tests/golden/sr_grad_golden.npz from tests/golden/make_golden.py --grad-only): the reference's own
    autograd under nn.L1Loss for the window-4, window-2 (x8) and window-16 presets of its CLI (src/main.py:218-219,286)."""
    import os
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "sr_grad_golden.npz"))
    cfg, sd, x, y = drct_case(sr_golden, name)
    m = build_train(cfg, sd, "fp32")
    xt = torch.from_numpy(x).cuda().requires_grad_(True)
    hr = torch.from_numpy(g[name + "/hr"]).cuda()
    out = m(xt)
    assert rel_err(out.detach().cpu().numpy(), y) < 2e-4
    loss = F.l1_loss(out, hr)
    assert abs(float(loss.detach()) - float(g[name + "/loss"])) < 1e-5 * abs(float(g[name + "/loss"]))
    loss.backward()
    assert rel_err(xt.grad.cpu().numpy(), g[name + "/grad_x"]) < 1e-3
    grads = dict(m.named_parameters())
    for k in [k for k in g.files if k.startswith(name + "/grad/")]:
        pname = k[len(name + "/grad/"):]
        e = rel_err(grads[pname].grad.cpu().numpy(), g[k])
        assert e < 1e-3, (pname, e)
    worst = 0.0
    for n, ref in zip([str(n) for n in g[name + "/grad_names"]], g[name + "/grad_l2"]):
        mine = float(grads[n].grad.double().pow(2).sum().sqrt())
        worst = max(worst, abs(mine - ref) / max(ref, 1e-12))
        assert abs(mine - ref) <= 1e-3 * max(ref, 1e-9), (n, mine, ref)
    print(name, "worst relative L2-norm error over the parameter tensors:", worst)


@pytest.mark.parametrize("gray", [True, False])
def test_training_forward_backward_with_droppath_matches_oracle(gray):
    from oracle import sr_ref as R
    from srad_amd import spec as S
    cfg = S.DRCTConfig(in_chans=1 if gray else 3, img_size=16, window_size=8, upscale=2 if gray else 4, n_rdg=2)
    sd = S.synth_state(S.drct_spec(cfg), seed=77, gain=1.0, cfg=cfg)
    B, H, W = 3, 16, 24
    x = S.synth_image("tr", (B, cfg.in_chans, H, W), seed=5)
    hr = S.synth_image("tr/hr", (B, cfg.in_chans, H * cfg.upscale, W * cfg.upscale), seed=6)
    gen = torch.Generator().manual_seed(1)
    keep = torch.floor(0.7 + torch.rand(2 * cfg.n_rdg * 5, B, generator=gen)) / 0.7        # independent per branch
    # oracle (torch CPU autograd)
    sdt = {k: torch.from_numpy(np.asarray(v)).clone().requires_grad_(np.asarray(v).dtype == np.float32) for k, v in sd.items()}
    keeps = [[(keep[2 * (i * 5 + k)], keep[2 * (i * 5 + k) + 1]) for k in range(5)] for i in range(cfg.n_rdg)]
    xr = torch.from_numpy(x).requires_grad_(True)
    ref = R.drct_forward(sdt, xr, cfg, keeps=keeps)
    F.l1_loss(ref, torch.from_numpy(hr)).backward()
    # engine
    m = build_train(cfg, sd, "fp32", drop_path_rate=0.1)
    m.keep_scale_override = keep.cuda()
    xt = torch.from_numpy(x).cuda().requires_grad_(True)
    out = m(xt)
    assert rel_err(out.detach().cpu().numpy(), ref.detach().numpy()) < 2e-4
    F.l1_loss(out, torch.from_numpy(hr).cuda()).backward()
    assert rel_err(xt.grad.cpu().numpy(), xr.grad.numpy()) < 1e-3
    worst = ("", 0.0)
    for n, p in m.named_parameters():
        e = rel_err(p.grad.cpu().numpy(), sdt[n].grad.numpy())
        if e > worst[1]:
            worst = (n, e)
        assert e < 1e-3, (n, e)
    print("worst parameter-gradient error:", worst)
    # a second backward accumulates (PyTorch semantics), zero_grad clears
    g1 = m.flat_grads.clone()
    out = m(xt)
    F.l1_loss(out, torch.from_numpy(hr).cuda()).backward()
    assert float((m.flat_grads - 2 * g1).abs().max()) <= 1e-4 * float(g1.abs().max())
    m.zero_grad()
    assert float(m.flat_grads.abs().max()) == 0.0


@pytest.mark.parametrize("prec", ["fp32", "bf16"])
@pytest.mark.parametrize("ws,img,up,B,H,W", [(2, 8, 4, 2, 8, 6), (4, 16, 4, 2, 16, 8), (16, 64, 4, 1, 32, 32), (2, 8, 8, 1, 8, 8)])
def test_training_other_window_sizes_matches_oracle_autograd(ws, img, up, B, H, W, prec):
    """The reference's other CLI presets (window_size = img_size // 4 in {2, 4, 16}; src/main.py:218-219,286 - x8 at 64 px builds
    window 2 as well): training forward + backward with DropPath masks against autograd of the oracle.  These take the unfused
    launches and the general attention backward (fp32 MFMAs); bf16 mode rounds the GEMM operands (bar as the ws-8 bf16 test)."""
    from oracle import sr_ref as R
    from srad_amd import spec as S
    cfg = S.DRCTConfig(in_chans=1, img_size=img, window_size=ws, upscale=up, n_rdg=2)
    sd = S.synth_state(S.drct_spec(cfg), seed=70 + ws, gain=1.0, cfg=cfg)
    x = S.synth_image("trws", (B, 1, H, W), seed=5)
    hr = S.synth_image("trws/hr", (B, 1, H * up, W * up), seed=6)
    gen = torch.Generator().manual_seed(2)
    keep = torch.floor(0.8 + torch.rand(2 * cfg.n_rdg * 5, B, generator=gen)) / 0.8
    sdt = {k: torch.from_numpy(np.asarray(v)).clone().requires_grad_(np.asarray(v).dtype == np.float32) for k, v in sd.items()}
    keeps = [[(keep[2 * (i * 5 + k)], keep[2 * (i * 5 + k) + 1]) for k in range(5)] for i in range(cfg.n_rdg)]
    xr = torch.from_numpy(x).requires_grad_(True)
    ref = R.drct_forward(sdt, xr, cfg, keeps=keeps)
    F.l1_loss(ref, torch.from_numpy(hr)).backward()
    m = build_train(cfg, sd, prec, drop_path_rate=0.1)
    assert m._can_train()
    m.keep_scale_override = keep.cuda()
    xt = torch.from_numpy(x).cuda().requires_grad_(True)
    out = m(xt)
    F.l1_loss(out, torch.from_numpy(hr).cuda()).backward()
    if prec == "fp32":
        assert rel_err(out.detach().cpu().numpy(), ref.detach().numpy()) < 2e-4
        assert rel_err(xt.grad.cpu().numpy(), xr.grad.numpy()) < 1e-3
        worst = ("", 0.0)
        for n, p in m.named_parameters():
            e = rel_err(p.grad.cpu().numpy(), sdt[n].grad.numpy())
            worst = max(worst, (n, e), key=lambda t: t[1])
            assert e < 1e-3, (n, e)
        print(f"window {ws}: worst parameter-gradient error {worst}")
    else:
        a = torch.cat([sdt[n].grad.reshape(-1) for n, _ in m.named_parameters()]).double()
        b = torch.cat([p.grad.reshape(-1).cpu() for _, p in m.named_parameters()]).double()
        cos = float((a * b).sum() / (a.norm() * b.norm()))
        print(f"window {ws} bf16: cosine(oracle fp32 grad, engine bf16 grad) = {cos}, norm ratio {float(b.norm() / a.norm())}")
        assert cos > 0.99 and 0.9 < float(b.norm() / a.norm()) < 1.1


def test_random_droppath_masks_have_reference_statistics():
    from srad_amd import spec as S
    cfg = S.DRCTConfig(in_chans=1, img_size=16, window_size=8, upscale=2, n_rdg=3)
    sd = S.synth_state(S.drct_spec(cfg), seed=3, gain=1.0, cfg=cfg)
    m = build_train(cfg, sd, "fp32", drop_path_rate=0.5)
    kp = m.drop_path_keep_probs()
    assert kp.shape == (15,) and float(kp[0]) == 1.0 and abs(float(kp[-1]) - (1 - 0.5 * 12 / 17)) < 1e-6
    x = torch.rand(64, 1, 8, 8, device="cuda") * 255
    with torch.enable_grad():
        m(x)
    k = m._keep.cpu()
    assert k.shape == (30, 64)
    assert torch.all(k[:10] == 1.0)                                 # RDG 0: rate 0 -> Identity
    vals = k[-1].unique()
    assert all(abs(float(v)) < 1e-6 or abs(float(v) - 1 / float(kp[-1])) < 1e-5 for v in vals)
    assert 0.3 < float((k[-10:] > 0).float().mean()) < 0.95          # keep_prob ~ 0.65


def test_fused_adam_training_reduces_loss_and_matches_torch_adam(sr_golden):
    from srad_amd.train import FusedAdam, train_step
    name = "drct_r2_rgb_x4"
    cfg, sd, x, y = drct_case(sr_golden, name)
    hr = torch.from_numpy(sr_golden[name + "/hr"]).cuda()
    xt = torch.from_numpy(x).cuda()
    m = build_train(cfg, sd, "fp32")
    opt = FusedAdam(m, lr=1e-4)
    # reference arithmetic: torch.optim.Adam over the same (view) parameters, driven by the same gradients
    m2 = build_train(cfg, sd, "fp32")
    opt2 = torch.optim.Adam(m2.parameters(), lr=1e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=0)
    losses = []
    for it in range(4):
        losses.append(float(train_step(m, xt, hr, opt)))
        # Adam normalises every gradient to +-lr, so noise-level differences in near-zero gradients (atomics
        # ordering) would flip whole steps: both optimizers see the SAME gradient buffer.
        m2.zero_grad()
        m2.flat_grads.copy_(m.flat_grads)
        opt2.step()
        diff = float((m.flat_params - m2.flat_params).abs().max())
        assert diff < 5e-6, (it, diff)
    print("losses", losses)
    assert losses[-1] < losses[0]
    # the updated weights are what the eval path uses
    m.eval()
    with torch.no_grad():
        e1 = m(xt)
    m2.eval()
    with torch.no_grad():
        e2 = m2(xt)
    assert rel_err(e1.cpu().numpy(), e2.cpu().numpy()) < 1e-4


def test_bf16_training_gradients_close_to_fp32(sr_golden):
    name = "drct_r2_rgb_x4"
    cfg, sd, x, y = drct_case(sr_golden, name)
    hr = torch.from_numpy(sr_golden[name + "/hr"]).cuda()
    xt = torch.from_numpy(x).cuda()
    gs = {}
    for prec in ("fp32", "bf16"):
        m = build_train(cfg, sd, prec)
        F.l1_loss(m(xt), hr).backward()
        gs[prec] = m.flat_grads.clone()
    a, b = gs["fp32"].double(), gs["bf16"].double()
    cos = float((a * b).sum() / (a.norm() * b.norm()))
    print("cosine(fp32 grad, bf16 grad) =", cos, " norm ratio", float(b.norm() / a.norm()))
    assert cos > 0.99 and 0.9 < float(b.norm() / a.norm()) < 1.1


def test_training_errors_are_reported():
    from srad_amd import spec as S
    from srad_amd.nets import DRCT
    cfg = S.DRCTConfig(in_chans=1, img_size=80, window_size=20, upscale=2, n_rdg=1)      # no CLI preset builds windows above 16
    o = Opt(cfg, "fp32")
    m = DRCT(o).cuda().train()
    assert not m._can_train()                                # the Trainer refuses this up front
    with pytest.raises(NotImplementedError, match="window sizes up to 16"):
        m(torch.zeros(1, 1, 20, 20, device="cuda"))
    with pytest.raises(RuntimeError, match="GPU only"):
        DRCT(o).train()(torch.zeros(1, 1, 8, 8))


def _write_grid_class(root, n_train=8, n_val=2, n_test=(3, 3), px=128):
    """A tiny MVTec-shaped tree: {root}/grid/{train,val}/good/{HR,LR_4}, {root}/grid/test/{good,bad}/{HR,LR_4}."""
    from PIL import Image
    rng = np.random.default_rng(0)
    yy, xx = np.mgrid[0:px, 0:px]

    def tile(i, defect=False):
        hr = 127 + 90 * np.sin(xx / (3.0 + 0.3 * i)) * np.cos(yy / (4.0 + 0.2 * i)) + rng.normal(0, 4, (px, px))
        if defect:
            hr[40:60, 50:80] += rng.normal(0, 60, (20, 30))
        return hr.clip(0, 255).astype(np.uint8)

    def put(d, i, defect=False):
        (d / "HR").mkdir(parents=True, exist_ok=True)
        (d / "LR_4").mkdir(parents=True, exist_ok=True)
        hr = tile(i, defect)
        Image.fromarray(hr).save(d / "HR" / f"{i:03d}.png")
        Image.fromarray(hr.reshape(px // 4, 4, px // 4, 4).mean((1, 3)).round().astype(np.uint8)).save(d / "LR_4" / f"{i:03d}.png")
    for i in range(n_train):
        put(root / "grid" / "train" / "good", i)
    for i in range(n_val):
        put(root / "grid" / "val" / "good", 100 + i)
    for i in range(n_test[0]):
        put(root / "grid" / "test" / "good", 200 + i)
    for i in range(n_test[1]):
        put(root / "grid" / "test" / "bad", 300 + i, defect=True)


def test_cli_train_writes_a_run_dir_the_evaluator_resolves_from_config_txt(tmp_path, capsys):
    """src/main.py train_drct end to end on the folder layout src/data.py reads, then ``evaluate --run-dir``:
    virtual epochs of 256 // batch steps (here test_every is cut to 3 for time), cosine schedule per epoch, Checkpoint
    files in the reference's format (config.txt, log.txt, model/*.pt, optimizer.pt, loss_log.pt, psnr_ssim_log.pt, result
    PNGs), and the evaluator finding model type / class / resolution / scale from config.txt ALONE (the directory is
    renamed so the name regex cannot help)."""
    from srad_amd import evaluate as E
    from srad_amd import main as Mn
    from srad_amd import options as Opt
    _write_grid_class(tmp_path / "data")
    args = Opt.parse_train_args(["--model-type", "drct", "--classe", "grid", "--resolution", "128", "--scale", "4", "--epochs", "2",
                                 "--batch-size", "4", "--data-root", str(tmp_path / "data"), "--save-dir", str(tmp_path / "exp")])
    opt = Mn.build_train_opt(args)
    assert opt.test_every == 64 and opt.print_every == 64 and opt.patch_size == 128 and opt.window_size == 8 and opt.loss == '1*L1'
    assert opt.lr == 1e-4 and (opt.beta1, opt.beta2, opt.epsilon, opt.weight_decay) == (0.9, 0.999, 1e-8, 0.0)
    opt.depths, opt.num_heads, opt.test_every, opt.print_every = (6, 6), (6, 6), 3, 1       # 2 RDG, 3 steps per epoch
    Mn.train_drct(opt)
    run = opt.save
    files = set(os.listdir(run))
    assert {"config.txt", "log.txt", "model", "results", "loss_log.pt", "psnr_ssim_log.pt", "optimizer.pt"} <= files
    assert sorted(os.listdir(os.path.join(run, "model"))) == ["model_best.pt", "model_latest.pt"]
    log = open(os.path.join(run, "log.txt")).read()
    assert "[Epoch 1]\tLearning rate: 1.00e-4" in log and "[Epoch 2]\tLearning rate: 5.0" in log      # cosine, T_max = 2: 5.005e-5
    assert log.count("[L1: ") == 6 and "[ x4]\tPSNR:" in log and "Total Training Time" in log
    loss_log = torch.load(os.path.join(run, "loss_log.pt"))
    assert tuple(loss_log.shape) == (2, 1) and float(loss_log[1, 0]) < float(loss_log[0, 0])
    pl = torch.load(os.path.join(run, "psnr_ssim_log.pt"))
    assert tuple(pl.shape) == (1, 2) and float(pl[0, 0]) > 5 and 0 < float(pl[0, 1]) <= 1
    assert sorted(os.listdir(os.path.join(run, "results", "x4"))) == ["100.png", "101.png"]
    osd = torch.load(os.path.join(run, "optimizer.pt"))
    assert osd["step"] == 6 and osd["exp_avg"].numel() > 1e6
    cfg = open(os.path.join(run, "config.txt")).read()
    assert "model_name: drct" in cfg and "classe: grid" in cfg and "patch_size: 128" in cfg and "upscale: 4" in cfg
    # the evaluator: everything from config.txt
    moved = str(tmp_path / "anonymous_run")
    os.rename(run.rstrip("/"), moved)
    inf = E.infer_from_run_dir(moved)
    assert (inf["model_type"], inf["classe"], inf["resolution"], inf["scale"]) == ("drct", "grid", 128, 4)
    # (the evaluator rebuilds the full-depth model from the option defaults; this run trained 2 RDGs, so the weights it
    # loads with strict=False cover the first two groups - what is checked here is the plumbing, not the AUC)
    capsys.readouterr()
    E.main(["--run-dir", moved, "--data-root", str(tmp_path / "data")])
    out = capsys.readouterr().out
    assert "Test AUCs - SSIM(best ws=" in out
    assert sorted(os.listdir(os.path.join(moved, "eval_results", "bad", "x4"))) == ["300.png", "301.png", "302.png"]


def test_torch_optimizer_and_load_state_dict_reach_the_engine(sr_golden):
    """After enable_training() the parameters are views into one flat buffer but keep their own version counters.  A
    torch optimizer step or load_state_dict must re-pack the engine's weights for the NEXT training forward and for eval
    (round-1 bug: only flat_params._version was watched, so the engine kept training on the initial weights)."""
    from srad_amd.train import FusedAdam, train_step
    name = "drct_r2_rgb_x4"
    cfg, sd, x, y = drct_case(sr_golden, name)
    hr = torch.from_numpy(sr_golden[name + "/hr"]).cuda()
    xt = torch.from_numpy(x).cuda()
    # reference trajectory: the fused Adam
    ma = build_train(cfg, sd, "fp32")
    oa = FusedAdam(ma, lr=1e-3)
    la = [float(train_step(ma, xt, hr, oa)) for _ in range(3)]
    # the reference loop, unchanged: model(x) -> F.l1_loss -> backward -> torch.optim.Adam.step
    mb = build_train(cfg, sd, "fp32")
    ob = torch.optim.Adam(mb.parameters(), lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0)
    lb = []
    for _ in range(3):
        ob.zero_grad(set_to_none=False)
        loss = F.l1_loss(mb(xt), hr)
        loss.backward()
        ob.step()
        lb.append(float(loss))
    assert lb[2] < lb[0], lb
    assert max(abs(a - b) / a for a, b in zip(la, lb)) < 1e-3, (la, lb)                  # same losses step by step
    assert float((ma.flat_params - mb.flat_params).abs().max()) < 5e-3 * 3                # <= lr per step apart (sign flips of ~0 grads)
    # eval after training uses the updated weights (both models)
    for m in (ma, mb):
        m.eval()
    with torch.no_grad():
        ea, eb = ma(xt), mb(xt)
    fresh = build_train(cfg, {k: v.detach().cpu().numpy() for k, v in mb.state_dict().items()}, "fp32").eval()
    with torch.no_grad():
        assert rel_err(eb.cpu().numpy(), fresh(xt).cpu().numpy()) < 1e-5
    assert rel_err(ea.cpu().numpy(), eb.cpu().numpy()) < 2e-2
    # load_state_dict after enable_training: the next training forward AND eval see the loaded weights
    mc = build_train(cfg, sd, "fp32")
    mc.load_state_dict(mb.state_dict())
    mc.train()
    out_c = mc(xt)
    mb.train()
    out_b = mb(xt)
    assert rel_err(out_c.detach().cpu().numpy(), out_b.detach().cpu().numpy()) < 1e-6
    # a parameter whose storage was swapped out is copied back into the flat buffer
    w = mc.get_parameter("conv_last.weight")
    w.data = torch.zeros_like(w.data)
    z = mc(xt)
    assert w.data_ptr() == mc._flat_homes[[p is w for p, _, _ in mc._flat_homes].index(True)][1]
    assert float(w.abs().max()) == 0.0 and not torch.equal(z, out_c)


def test_eval_after_graph_replays_uses_the_stepped_weights(sr_golden):
    """GraphedTrainStep replays re-pack at the START of the captured step, so after a replay the packs are one Adam step
    behind the flat parameters: eval must re-pack (round-1 bug: Trainer.test scored stale weights from epoch 2 on)."""
    from srad_amd.train import FusedAdam, GraphedTrainStep
    name = "drct_r2_rgb_x4"
    cfg, sd, x, y = drct_case(sr_golden, name)
    hr = torch.from_numpy(sr_golden[name + "/hr"]).cuda()
    xt = torch.from_numpy(x).cuda()
    m = build_train(cfg, sd, "fp32")
    step = GraphedTrainStep(m, FusedAdam(m, lr=1e-3), warmup=2)
    for _ in range(5):
        step(xt, hr)
    assert step._graphs, "the step was not captured"
    m.eval()
    with torch.no_grad():
        got = m(xt)
    fresh = build_train(cfg, {k: v.detach().cpu().numpy() for k, v in m.state_dict().items()}, "fp32").eval()
    with torch.no_grad():
        assert rel_err(got.cpu().numpy(), fresh(xt).cpu().numpy()) < 1e-5


def _dp_worker(rank, world, port, cfg_tuple, seed, x, hr, q):
    import os
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from srad_amd import spec as S
    from srad_amd.train import FusedAdam, GradReducer, train_step
    cfg = S.DRCTConfig(*cfg_tuple)
    sd = S.synth_state(S.drct_spec(cfg), seed=seed, gain=1.0, cfg=cfg)
    m = build_train(cfg, sd, "fp32")
    opt = FusedAdam(m, lr=1e-4)
    red = GradReducer().attach(m)
    xs, hs = torch.from_numpy(x[rank::world]).cuda(), torch.from_numpy(hr[rank::world]).cuda()
    loss = train_step(m, xs, hs, opt, red)
    torch.cuda.synchronize()
    q.put((rank, float(loss), m.flat_grads.cpu().numpy(), m.flat_params.cpu().numpy()))
    dist.barrier()
    dist.destroy_process_group()


def test_data_parallel_step_two_ranks_equals_one_rank_on_the_full_batch():
    """Two processes (one GPU, gloo) each take half of a 4-image batch; the bucket hooks all-reduce the gradients while
    the backward runs.  Summed gradients / world and the updated weights must equal the single-process step on the
    whole batch (L1 'mean' over the batch = mean of the per-shard means)."""
    import socket
    import torch.multiprocessing as mp
    from srad_amd import spec as S
    from srad_amd.train import FusedAdam, train_step
    cfg_tuple = (1, 16, 8, 2, 180, 2)        # in_chans, img_size, window, upscale, embed_dim, n_rdg
    cfg = S.DRCTConfig(*cfg_tuple)
    x = S.synth_image("dp", (4, 1, 16, 16), seed=5)
    hr = S.synth_image("dp/hr", (4, 1, 32, 32), seed=6)
    sd = S.synth_state(S.drct_spec(cfg), seed=9, gain=1.0, cfg=cfg)
    m = build_train(cfg, sd, "fp32")
    opt = FusedAdam(m, lr=1e-4)
    loss = float(train_step(m, torch.from_numpy(x).cuda(), torch.from_numpy(hr).cuda(), opt))
    g_ref, p_ref = m.flat_grads.cpu().numpy(), m.flat_params.cpu().numpy()
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_dp_worker, args=(r, 2, port, cfg_tuple, 9, x, hr, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = {}
    for _ in procs:
        r, l, g, pp = q.get(timeout=300)
        res[r] = (l, g, pp)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert abs(0.5 * (res[0][0] + res[1][0]) - loss) < 1e-4 * abs(loss)
    assert np.array_equal(res[0][1], res[1][1])                       # both ranks hold the same summed gradients
    gmax = np.abs(g_ref).max()
    assert np.abs(res[0][1] * 0.5 - g_ref).max() < 2e-4 * gmax
    # Adam turns noise-level gradients into +-lr steps; compare where the gradient is clearly non-zero
    big = np.abs(g_ref) > 1e-3 * gmax
    assert np.abs(res[0][2] - p_ref)[big].max() < 2e-5


def test_bf16_training_fused_forward_matches_unfused(sr_golden, monkeypatch):
    """bf16 mode: the training forward through the two fused block kernels (with DropPath and the saved activations
    written from inside them) against the eight-launch path (SRAD_NO_FUSE), same DropPath masks: outputs and
    gradients agree to bf16 rounding."""
    name = "drct_r2_rgb_x4"
    cfg, sd, x, y = drct_case(sr_golden, name)
    hr = torch.from_numpy(sr_golden[name + "/hr"]).cuda()
    xt = torch.from_numpy(x).cuda()
    gen = torch.Generator().manual_seed(3)
    keep = (torch.floor(0.8 + torch.rand(2 * cfg.n_rdg * 5, x.shape[0], generator=gen)) / 0.8).cuda()
    outs, grads = {}, {}
    for mode in ("fused", "unfused"):
        if mode == "unfused":
            monkeypatch.setenv("SRAD_NO_FUSE", "1")
        else:
            monkeypatch.delenv("SRAD_NO_FUSE", raising=False)
        m = build_train(cfg, sd, "bf16", drop_path_rate=0.1)
        m.keep_scale_override = keep
        out = m(xt)
        F.l1_loss(out, hr).backward()
        outs[mode], grads[mode] = out.detach().clone(), m.flat_grads.clone()
    rng = float(outs["unfused"].max() - outs["unfused"].min())
    assert float((outs["fused"] - outs["unfused"]).abs().max()) / rng < 1e-2
    a, b = grads["fused"].double(), grads["unfused"].double()
    cos = float((a * b).sum() / (a.norm() * b.norm()))
    print("fused vs unfused bf16 training: cosine of the gradients", cos)
    assert cos > 0.999


@pytest.mark.parametrize("batch", [2, 8])
def test_bf16_fused_mlp_backward_matches_unfused(sr_golden, monkeypatch, batch):
    """bf16 mode: the fused backward kernels - MLP branch (both data gradients + LayerNorm2 backward) and the qkv data
    gradient + LayerNorm1 backward, 16-row tiles below 8192 tokens, 32-row tiles from there - against the separate
    GEMM / LayerNorm launches, same forward, same DropPath masks."""
    name = "drct_r2_rgb_x4"
    cfg, sd, x, y = drct_case(sr_golden, name)
    reps = batch // x.shape[0]
    xt = torch.from_numpy(x).cuda().repeat(reps, 1, 1, 1)
    hr = torch.from_numpy(sr_golden[name + "/hr"]).cuda().repeat(reps, 1, 1, 1)
    hr = hr + torch.linspace(0, 20, batch, device="cuda").view(-1, 1, 1, 1)      # the copies see different gradients
    gen = torch.Generator().manual_seed(5)
    keep = (torch.floor(0.8 + torch.rand(2 * cfg.n_rdg * 5, batch, generator=gen)) / 0.8).cuda()
    grads = {}
    for mode in ("fused", "unfused"):
        monkeypatch.delenv("SRAD_NO_FUSE", raising=False)
        m = build_train(cfg, sd, "bf16", drop_path_rate=0.1)
        m.keep_scale_override = keep
        if mode == "unfused":
            # after the engine is built (its forward stays fused), before the forward (which saves q | k | v in the form the
            # backward will take): only the backward changes
            monkeypatch.setenv("SRAD_NO_FUSE", "1")
        out = m(xt)
        F.l1_loss(out, hr).backward()
        torch.cuda.synchronize()
        grads[mode] = m.flat_grads.clone()
        named = {n: p.grad.clone() for n, p in m.named_parameters()}
        grads[mode + "_named"] = named
    monkeypatch.delenv("SRAD_NO_FUSE", raising=False)
    a, b = grads["fused"].double(), grads["unfused"].double()
    cos = float((a * b).sum() / (a.norm() * b.norm()))
    worst = 0.0
    for n, g in grads["unfused_named"].items():
        if "norm" in n or "mlp.fc" in n or "attn" in n or "adjust" in n:
            worst = max(worst, float((grads["fused_named"][n] - g).norm() / g.norm().clamp_min(1e-30)))
    print(f"fused vs unfused MLP backward (batch {batch}): cosine {cos:.7f}, worst relative L2 over norm / fc / qkv / adjust tensors {worst:.2e}")
    assert cos > 0.9999 and worst < 2e-2


@pytest.mark.parametrize("prec", ["bf16", "fp32"])
def test_graphed_train_step_matches_eager(sr_golden, prec):
    """The whole training step as one replayed hipGraph (two eager warm-up steps, then capture + replays) walks the
    same parameter trajectory as the eager ``train_step``: same kernels, same fixed-order reductions -> same bits;
    Adam's step count / learning rate reach the replays through device memory."""
    from srad_amd.train import FusedAdam, GraphedTrainStep, train_step
    name = "drct_r2_rgb_x4"
    cfg, sd, x, y = drct_case(sr_golden, name)
    xt = torch.from_numpy(x).cuda()
    hr = torch.from_numpy(sr_golden[name + "/hr"]).cuda()
    runs = {}
    for mode in ("eager", "graph"):
        m = build_train(cfg, sd, prec)
        opt = FusedAdam(m, lr=1e-3)
        step = GraphedTrainStep(m, opt, warmup=2) if mode == "graph" else (lambda a, b: train_step(m, a, b, opt))
        losses = []
        for i in range(6):
            if i == 4:
                opt.param_groups[0]["lr"] = 5e-4                   # a scheduler step between replays
            losses.append(step(xt + i, hr))                        # inputs change per step too
        torch.cuda.synchronize()
        runs[mode] = (torch.stack([l.double() for l in losses]).cpu(), m.flat_params.clone(), opt.step_count)
        if mode == "graph":
            assert len(step._graphs) == 1
    assert runs["graph"][2] == runs["eager"][2] == 6
    assert torch.equal(runs["graph"][0], runs["eager"][0]), (runs["graph"][0], runs["eager"][0])
    assert torch.equal(runs["graph"][1], runs["eager"][1])


def test_graphed_train_step_draws_new_droppath_masks(sr_golden):
    """DropPath inside the captured step: every replay draws new per-sample masks (graph-safe Philox offset)."""
    from srad_amd.train import FusedAdam, GraphedTrainStep
    name = "drct_r2_rgb_x4"
    cfg, sd, x, y = drct_case(sr_golden, name)
    xt = torch.from_numpy(x).cuda().repeat(4, 1, 1, 1)
    hr = torch.from_numpy(sr_golden[name + "/hr"]).cuda().repeat(4, 1, 1, 1)
    m = build_train(cfg, sd, "bf16", drop_path_rate=0.5)
    opt = FusedAdam(m, lr=0.0)                                     # parameters stay put: only the masks change the loss
    step = GraphedTrainStep(m, opt, warmup=1)
    masks, losses = [], []
    for i in range(5):
        losses.append(float(step(xt, hr)))
        masks.append(m._keep.clone())
    assert len(step._graphs) == 1
    assert len({tuple(k.flatten().tolist()) for k in masks[1:]}) >= 3          # replays 2..5: different masks
    assert len(set(losses[1:])) >= 3
    for k in masks:                                                # floor(keep + U) / keep: 0 or 1 / keep_prob of the RDG
        assert bool(((k == 0) | (k >= 1.0)).all()) and bool((k[:10] == 1.0).all())      # RDG 0: rate 0 (drct.py:819)
