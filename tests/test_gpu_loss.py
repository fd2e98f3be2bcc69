"""GPU: the loss types of the reference's loss factory (src/loss.py) on the engine's reductions (C ABI srad_loss_forward /
srad_loss_backward) against
  * the reference's own values and autograd gradients (tests/golden/host_golden.npz, written from the imported src/loss.py);
  * autograd of the oracle (oracle/host_ref.py) on other shapes;
and the Loss object's log / value_and_grad / autograd paths, the tensor-wise Adam of the dual models, and the bit-equality of
the generic L1 path with the training step's dedicated L1 kernels.  Bars: values 1e-5 relative, gradients 1e-4 of the gradient's
max (fp32 reductions in another order; SSIM's 11 x 11 means are accumulated in a different order than torch's conv2d)."""
import numpy as np
import pytest
import torch

from oracle import host_ref as O

pytestmark = pytest.mark.gpu

SPECS = ["1*L1", "1*MSE", "1*PSNR", "1*SSIM", "0.7*L1+0.3*SSIM", "1*MSE+0.05*PSNR"]


class A:
    rgb_range, batch_size = 255, 3

    def __init__(self, loss):
        self.loss = loss


def _key(spec):
    return spec.replace("*", "x").replace("+", "_")


@pytest.mark.parametrize("tag", ["gray48", "rgb40", "gray16", "gray_crop"])
def test_loss_types_match_reference_goldens(host_golden, tag):
    from srad_amd.loss import Loss
    g = host_golden
    sr, hr = torch.from_numpy(g[f"loss/{tag}/sr"]).cuda(), torch.from_numpy(g[f"loss/{tag}/hr"]).cuda()
    for spec in (["1*SSIM"] if tag == "gray_crop" else SPECS):
        ref_v, ref_g = float(g[f"loss/{tag}/{_key(spec)}/value"]), g[f"loss/{tag}/{_key(spec)}/grad"]
        lf = Loss(A(spec))
        lf.start_log()
        # the reference's call: value with autograd
        x = sr.clone().requires_grad_(True)
        val = lf(x, hr)
        val.backward()
        assert abs(float(val) - ref_v) <= 1e-5 * max(1.0, abs(ref_v)), (tag, spec, float(val), ref_v)
        e = np.abs(x.grad.cpu().numpy() - ref_g).max() / np.abs(ref_g).max()
        assert e < 1e-4, (tag, spec, e)
        # the fused step's call: no autograd, same numbers
        v2, dy = lf.value_and_grad(sr, hr)
        assert abs(float(v2) - float(val)) <= 1e-6 * max(1.0, abs(ref_v)) and torch.allclose(dy, x.grad, rtol=1e-5, atol=1e-9 + 1e-6 * float(x.grad.abs().max()))
        lf.end_log(2)                                                       # two evaluations were logged
        assert np.allclose(lf.log.numpy(), g[f"loss/{tag}/{_key(spec)}/log"], rtol=2e-5, atol=1e-6), (lf.log, g[f"loss/{tag}/{_key(spec)}/log"])


@pytest.mark.parametrize("spec", ["1*SSIM", "0.5*PSNR+2*L1"])
def test_loss_types_match_oracle_autograd_on_other_shapes(spec):
    from srad_amd.loss import Loss
    g = torch.Generator().manual_seed(9)
    for (B, C, H, W, extra) in [(3, 3, 64, 48, 0), (2, 1, 21, 30, 0), (1, 3, 24, 24, 8)]:
        if extra and "SSIM" not in spec:
            continue
        hr = torch.rand(B, C, H, W, generator=g) * 300 - 20
        sr = (hr[0:1].repeat(B, 1, 1, 1) if False else hr) + torch.randn(B, C, H, W, generator=g) * 20
        if extra:
            sr = torch.nn.functional.pad(sr, (0, extra, 0, extra), value=3.0)
        x = sr.clone().requires_grad_(True)
        ref = O.total_loss(spec, x, hr, batch_size=3)
        ref.backward()
        lf = Loss(A(spec))
        lf.start_log()
        v, dy = lf.value_and_grad(sr.cuda(), hr.cuda())
        assert abs(float(v) - float(ref.detach())) <= 1e-5 * max(1.0, abs(float(ref.detach())))
        assert float((dy.cpu() - x.grad).abs().max()) <= 1e-4 * float(x.grad.abs().max()) + 1e-12
        if extra:
            assert float(dy[..., H:, :].abs().max()) == 0 and float(dy[..., :, W:].abs().max()) == 0   # the cropped-away margin


def test_generic_l1_equals_the_training_steps_l1_kernels():
    from srad_amd import _lib as L
    from srad_amd import metrics as M
    from srad_amd.loss import Loss
    g = torch.Generator().manual_seed(2)
    sr, hr = (torch.rand(2, 1, 128, 128, generator=g) * 255).cuda(), (torch.rand(2, 1, 128, 128, generator=g) * 255).cuda()
    sr[0, 0, 0, :5] = hr[0, 0, 0, :5]                                  # exact ties: sign(0) = 0
    v, dy = Loss(A("1*L1")).value_and_grad(sr, hr)
    ref = torch.empty_like(sr)
    L.check(L.lib().srad_l1_grad(L.dptr(sr), L.dptr(hr), L.dptr(ref), sr.numel(), 1.0 / sr.numel(), L.current_stream_ptr()), "l1_grad")
    assert torch.equal(dy, ref) and float(dy[0, 0, 0, :5].abs().max()) == 0
    assert abs(float(v) - float(M.l1_loss(sr, hr))) < 1e-9 * float(v)
    assert abs(float(v) - float(torch.nn.functional.l1_loss(sr, hr))) < 1e-5 * float(v)


def test_loss_refuses_cpu_tensors_and_unknown_types():
    from srad_amd.loss import Loss
    with pytest.raises(RuntimeError, match="GPU only"):
        Loss(A("1*L1"))(torch.zeros(1, 1, 8, 8), torch.zeros(1, 1, 8, 8))
    with pytest.raises(AssertionError, match="Unsupported loss type"):
        Loss(A("1*VGG"))
    with pytest.raises(ValueError, match="differ"):
        Loss(A("1*MSE"))(torch.zeros(1, 1, 8, 8, device="cuda"), torch.zeros(1, 1, 8, 9, device="cuda"))


def test_tensor_adam_matches_torch_adam():
    """The dual models' optimizer (src/trainer.py:62-73) is the engine's Adam kernel applied per tensor."""
    from srad_amd.train import TensorAdam
    g = torch.Generator().manual_seed(4)
    ps = [torch.nn.Parameter(torch.randn(20, 3, 3, 3, generator=g).cuda()), torch.nn.Parameter(torch.randn(3, 20, 3, 3, generator=g).cuda())]
    qs = [torch.nn.Parameter(p.detach().clone()) for p in ps]
    a = TensorAdam(ps, lr=1e-3, weight_decay=1e-8)
    b = torch.optim.Adam(qs, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-8)
    for it in range(4):
        for p, q in zip(ps, qs):
            gr = torch.randn(p.shape, generator=g).cuda() * (0.1 + it)
            p.grad, q.grad = gr.clone(), gr.clone()
        a.step()
        b.step()
        if it == 1:
            a.param_groups[0]["lr"] = b.param_groups[0]["lr"] = 5e-4
    for p, q in zip(ps, qs):
        assert float((p - q).abs().max()) < 2e-6
    sd = a.state_dict()
    assert sd["step"] == 4 and len(sd["exp_avg"]) == 2
    a.zero_grad()
    assert ps[0].grad is None


def test_rccl_allreduce_of_the_gradient_buckets_world_one():
    """The data-parallel step over the ``nccl`` backend (= RCCL): a one-rank process group on this box's single GPU runs
    the real bucket hooks, side stream and collectives (a world of one is what one GPU allows; the N-rank launch is
    bench.py --gpus N / main.py --gpus N).  Summing over one rank must leave the single-GPU step unchanged."""
    import socket
    import torch.distributed as dist
    from srad_amd import spec as S
    from srad_amd.train import FusedAdam, GradReducer, train_step
    from tests.test_gpu_train import build_train
    cfg = S.DRCTConfig(1, 16, 8, 2, 180, 2)
    sd = S.synth_state(S.drct_spec(cfg), seed=9, gain=1.0, cfg=cfg)
    x = torch.from_numpy(S.synth_image("dp", (2, 1, 16, 16), seed=5)).cuda()
    hr = torch.from_numpy(S.synth_image("dp/hr", (2, 1, 32, 32), seed=6)).cuda()
    m0 = build_train(cfg, sd, "fp32")
    l0 = float(train_step(m0, x, hr, FusedAdam(m0, lr=1e-4)))
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        assert dist.get_backend() == "nccl"
        m1 = build_train(cfg, sd, "fp32")
        red = GradReducer()
        red.world = 2                                   # force the hooks on: attach() skips them for a world of one
        red.attach(m1)
        red.world = 1
        assert m1.on_bucket is not None
        l1 = float(train_step(m1, x, hr, FusedAdam(m1, lr=1e-4), red))
        torch.cuda.synchronize()
        assert red.comm_stream is not None              # the collectives ran on the side stream
        assert l1 == l0 and torch.equal(m1.flat_grads, m0.flat_grads) and torch.equal(m1.flat_params, m0.flat_params)
    finally:
        dist.destroy_process_group()
