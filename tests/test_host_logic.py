"""CPU: host-side logic that mirrors the reference's CLI / option / run-dir handling, and the
multi-process (world_size 2, gloo) sharding + gather of the image-parallel evaluator."""
import os
import socket

import numpy as np
import pytest
import torch

from srad_amd import evaluate as E
from srad_amd import options as Opt


def test_option_objects_follow_reference_rules():
    o = Opt.build_opt('drct', 'grid', 128, 4)
    assert (o.window_size, o.img_size, o.upscale, o.scale, o.n_colors) == (8, 32, 4, [4], 1)
    assert o.embed_dim == 180 and len(o.depths) == 12 and o.mlp_ratio == 2 and o.loss == '1*L1'
    o = Opt.build_opt('drct', 'carpet', 1024, 4)
    assert (o.window_size, o.img_size, o.n_colors) == (64, 256, 3)            # window = img_size // 4 (C5)
    assert [(Opt.build_opt('drn-l', 'grid', 64, s).n_blocks, Opt.build_opt('drn-l', 'grid', 64, s).n_feats,
             Opt.build_opt('drn-l', 'grid', 64, s).scale) for s in (2, 4, 8)] == [(44, 40, [2]), (40, 20, [2, 4]), (36, 10, [2, 4, 8])]
    a = Opt.parse_train_args(['--model-type', 'drn-l', '--scale', '2', '--resolution', '64', '--device', 'cpu', '--lr', '0.5'])
    assert (a.model_type, a.scale, a.resolution, a.batch_size, a.epochs) == ('drn-l', 2, 64, 4, 2)
    with pytest.raises(SystemExit):
        Opt.parse_train_args(['--scale', '3'])
    e = Opt.parse_eval_args(['--run-dir', 'x', '--dtype', 'bf16'])
    assert e.run_dir == 'x' and e.dtype == 'bf16' and e.batch_size == 1


def test_config_file_sets_defaults(tmp_path):
    cfg = tmp_path / "c.yaml"
    cfg.write_text("model-type: drn-l\nbatch-size: 7\n")
    a = Opt.parse_train_args(['--config', str(cfg)])
    assert a.model_type == 'drn-l' and a.batch_size == 7
    a = Opt.parse_train_args(['--config', str(cfg), '--batch-size', '3'])
    assert a.batch_size == 3


def test_infer_from_run_dir(tmp_path):
    run = tmp_path / "experiment" / "drn-l" / "mvtec_carpet_256_X412:30:59"
    run.mkdir(parents=True)
    inf = E.infer_from_run_dir(str(run))
    assert inf['model_type'] == 'drn-l' and inf['classe'] == 'carpet' and inf['resolution'] == 256
    assert inf['scale'] == 412            # the reference's regex swallows the HH of the timestamp ...
    (run / "config.txt").write_text("2025-01-01\n\nmodel_name: drn-l\nscale: [2, 4]\npatch_size: 256\nclasse: carpet\ndataset: mvtec\n")
    assert E.infer_from_run_dir(str(run))['scale'] == 4        # ... and config.txt puts it right
    with pytest.raises(FileNotFoundError):
        E.resolve_checkpoint(type("A", (), {"checkpoint": "", "run_dir": str(run)})())
    (run / "model").mkdir()
    (run / "model" / "model_latest.pt").write_bytes(b"x")
    assert E.resolve_checkpoint(type("A", (), {"checkpoint": "", "run_dir": str(run)})()).endswith("model_latest.pt")


def test_cpu_device_is_refused_not_emulated():
    with pytest.raises(SystemExit, match="no CPU fallback"):
        E.main(['--device', 'cpu', '--checkpoint', 'nope.pt'])
    from srad_amd.model import Model
    o = Opt.build_opt('drct', 'grid', 128, 4)
    o.cpu = True
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        Model(o)


def _worker(rank, world, port, n, q):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    mine = E.shard_indices(n, rank, world)
    rows = np.stack([np.array([i, i * i, -i], dtype=np.float64) for i in mine]) if mine else np.zeros((0, 3))
    full = E.gather_score_rows(mine, rows, n, rank, world)
    q.put((rank, mine, None if full is None else full.tolist()))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n", [7, 78])
def test_image_parallel_sharding_gloo_world2(n):
    """world_size 2 over gloo: every image is scored by exactly one rank and rank 0 rebuilds the table."""
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = {}
    for _ in procs:
        r, mine, full = q.get(timeout=120)
        res[r] = (mine, full)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sorted(res[0][0] + res[1][0]) == list(range(n)) and not set(res[0][0]) & set(res[1][0])
    assert res[1][1] is None
    assert res[0][1] == [[float(i), float(i * i), float(-i)] for i in range(n)]


# ----------------------------------------------------------------------------------- data-parallel training (C4)
class _FakeModel:
    """What GradReducer needs of the engine module: a flat gradient buffer and its bucket table."""

    def __init__(self, n, buckets):
        self.flat_grads = torch.zeros(n)
        self.grad_buckets = buckets
        self.on_bucket = None


def _ddp_worker(rank, world, port, q):
    import torch.distributed as dist
    from srad_amd.train import GradReducer
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    buckets = [(700, 300), (300, 400), (0, 300), (1000, 0)]          # completion order, incl. an empty bucket
    m = _FakeModel(1000, buckets)
    red = GradReducer(overlap=True).attach(m)
    assert m.on_bucket is not None and red.grad_scale == 1.0 / world
    g = torch.Generator().manual_seed(rank)
    local = torch.randn(1000, generator=g)
    m.flat_grads.copy_(local)
    for b in range(len(buckets)):            # what srad_drct_backward's hook does, bucket by bucket
        m.on_bucket(b)
    red.finish()
    hooked = m.flat_grads.clone()
    m.flat_grads.copy_(local)
    red.reduce_all(m.flat_grads, buckets)   # the non-overlapped form
    q.put((rank, local.tolist(), hooked.tolist(), m.flat_grads.tolist()))
    dist.barrier()
    dist.destroy_process_group()


def test_gradient_buckets_allreduce_gloo_world2():
    """world_size 2 over gloo: after the bucket hooks every rank holds the SUM of both ranks' gradients in every
    bucket (the 1/world goes into the optimizer's grad_scale), and the hook path equals the plain path."""
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_ddp_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = {}
    for _ in procs:
        r, local, hooked, plain = q.get(timeout=120)
        res[r] = (np.array(local), np.array(hooked), np.array(plain))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    total = res[0][0] + res[1][0]
    for r in (0, 1):
        assert np.allclose(res[r][1], total, atol=1e-6)
        assert np.array_equal(res[r][1], res[r][2])


class _FlatNet(torch.nn.Module):
    """A CPU stand-in with the engine modules' training surface: parameters and gradients are views of two flat buffers."""

    def __init__(self, seed):
        super().__init__()
        g = torch.Generator().manual_seed(seed)
        shapes = [(2, 1, 3, 3), (2,), (1, 2, 3, 3)]
        n = sum(int(np.prod(s)) for s in shapes)
        self.flat_params = torch.randn(n, generator=g) * 0.3
        self.flat_grads = torch.zeros(n)
        self.ps, off = torch.nn.ParameterList(), 0
        offs = []
        for s in shapes:
            k = int(np.prod(s))
            p = torch.nn.Parameter(self.flat_params[off:off + k].view(s))
            p.grad = self.flat_grads[off:off + k].view(s)
            self.ps.append(p)
            offs.append(off)
            off += k
        # the engine's bucket surface (nets.DRN.enable_training): ranges in completion order + a hook fired when a bucket is final
        self.grad_buckets = [(offs[2], n - offs[2]), (0, offs[2])]
        self.on_bucket, self.fired = None, []
        pending = {0: 1, 1: 2}

        def done(b):
            def hook(_p):
                pending[b] -= 1
                if pending[b] == 0:
                    pending[b] = 1 if b == 0 else 2
                    self.fired.append(b)
                    if self.on_bucket is not None:
                        self.on_bucket(b)
            return hook
        self.ps[2].register_post_accumulate_grad_hook(done(0))
        self.ps[0].register_post_accumulate_grad_hook(done(1))
        self.ps[1].register_post_accumulate_grad_hook(done(1))

    def forward(self, x):
        import torch.nn.functional as F
        y = F.conv2d(F.relu(F.conv2d(x, self.ps[0], self.ps[1], padding=1)), self.ps[2], padding=1)
        return [y, F.interpolate(y, scale_factor=2.0)]


class _Sgd:
    def __init__(self, params):
        self.params, self.scales = list(params), []

    def zero_grad(self):
        for p in self.params:
            if p.grad is not None:
                p.grad.zero_()

    def step(self, grad_scale=1.0):
        self.scales.append(grad_scale)
        with torch.no_grad():
            for p in self.params:
                p -= 0.1 * grad_scale * p.grad


def _drn_dp_worker(rank, world, port, q):
    import torch.distributed as dist
    from srad_amd.loss import Loss
    from srad_amd.train import GradReducer, drn_train_step
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.manual_seed(0)
    net = _FlatNet(1)                                              # identical replicas
    dual = torch.nn.Conv2d(1, 1, 3, stride=2, padding=1, bias=False)
    with torch.no_grad():
        dual.weight.fill_(0.1)
    red = GradReducer(overlap=False)
    g = torch.Generator().manual_seed(5)
    lr_all, hr_all = torch.rand(4, 1, 8, 8, generator=g), torch.rand(4, 1, 16, 16, generator=g)
    sl = slice(rank, None, world) if world > 1 else slice(None)
    opt, dopt = _Sgd(net.parameters()), _Sgd(dual.parameters())
    loss = drn_train_step(net, [dual], [lr_all[sl]], hr_all[sl], opt, [dopt], 0.1, red if world > 1 else None,
                          torch.nn.functional.l1_loss)            # (the engine's L1 reduction is GPU-only; the step's plumbing is what runs here)
    # the logging Loss: every rank notes its slice's loss, the flushed row is the global minibatch's mean
    class A: loss, rgb_range, batch_size, print_every, save = "1*L1", 255, 2, 1, "."
    L_ = Loss.__new__(Loss)
    L_.loss, L_.log, L_._acc = [{"type": "L1", "weight": 1.0, "function": None}], torch.zeros(1, 1), None
    L_.note([loss])
    L_._flush()
    assert net.fired == [0, 1] and (world == 1 or red.model is net)       # world 2: reduced bucket by bucket from the hook
    q.put((rank, net.flat_params.tolist(), dual.weight.detach().reshape(-1).tolist(), float(loss), float(L_.log[-1, 0]), opt.scales))
    dist.barrier()
    dist.destroy_process_group()


def test_drn_data_parallel_step_and_loss_log_gloo_world2():
    """``drn_train_step``'s reducer branch (the SR net's gradient buckets all-reduced from the backward's hook as they complete,
    the dual models' gradients after it, 1 / world into
    the optimizers) and ``Loss._flush`` under world 2 over gloo: two ranks on half batches end with the parameters of one rank
    on the full batch, and the logged loss is the mean over the ranks."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    out = {}
    for world in (1, 2):
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = s.getsockname()[1]
        q = ctx.Queue()
        procs = [ctx.Process(target=_drn_dp_worker, args=(r, world, port, q)) for r in range(world)]
        for p in procs:
            p.start()
        res = [q.get(timeout=180) for _ in procs]
        for p in procs:
            p.join(timeout=60)
            assert p.exitcode == 0
        out[world] = sorted(res)
    one = out[1][0]
    for r in (0, 1):
        rank, params, dw, loss, logged, scales = out[2][r]
        assert np.allclose(params, one[1], atol=1e-6) and np.allclose(dw, one[2], atol=1e-6)       # half batches, summed, / 2 == the full batch
        assert scales == [0.5]
        assert abs(logged - 0.5 * (out[2][0][3] + out[2][1][3])) < 1e-6                           # the log row: mean over the ranks
    assert abs(out[2][0][4] - one[4]) < 1e-5                                                        # == the one-rank log of the full batch (L1 is a mean)


@pytest.mark.parametrize("scale,n_feats", [(2, 40), (4, 20), (8, 10)])
def test_drn_gradient_buckets_tile_the_flat_buffer(scale, n_feats):
    """Host-only calls of the C ABI (no GPU): the DRN engine's gradient buckets (srad_drn_bucket_range, the data-parallel
    all-reduce units of srad_drn_backward's hook) are phase + 2 contiguous ranges that tile the flat gradient buffer, in
    completion order: tails, the up phases finest first, then everything in front of them; every parameter's offset falls
    into exactly one bucket.  Incl. the x8 preset, whose padded layers keep their REAL sizes in the flat buffer."""
    import ctypes as C
    from srad_amd import _lib as L
    cfg = L.DrnConfig(1, scale, 2, n_feats, 0.2, 255.0, L.PREC_F32, 0)
    h = C.c_void_p()
    L.check(L.lib().srad_drn_create(C.byref(cfg), C.byref(h)), "drn_create")
    try:
        total = C.c_int64()
        L.check(L.lib().srad_drn_train_param_floats(h, C.byref(total)), "train_param_floats")
        nb = L.lib().srad_drn_num_buckets(h)
        phase = {2: 1, 4: 2, 8: 3}[scale]
        assert nb == phase + 2
        a, n = C.c_int64(), C.c_int64()
        spans = []
        for b in range(nb):
            L.check(L.lib().srad_drn_bucket_range(h, b, C.byref(a), C.byref(n)), "drn_bucket_range")
            spans.append((a.value, n.value))
        assert all(n > 0 for _, n in spans)
        assert spans[-1][0] == 0                                              # the last bucket starts the buffer ...
        order = [spans[-1]] + spans[-2:0:-1] + [spans[0]]                      # ... then the up phases coarse -> fine, then the tails
        assert all(x[0] + x[1] == y[0] for x, y in zip(order, order[1:])) and sum(order[-1]) == total.value
        off, numel = C.c_int64(), C.c_int64()
        name = C.c_char_p()
        tails = 0
        for i in range(L.lib().srad_drn_num_params(h)):
            L.check(L.lib().srad_drn_param_info(h, i, C.byref(name), C.byref(numel)), "param_info")
            L.check(L.lib().srad_drn_train_param_offset(h, i, C.byref(off)), "param_offset")
            inside = [b for b, (s0, n0) in enumerate(spans) if s0 <= off.value and off.value + numel.value <= s0 + n0]
            assert len(inside) == 1, name.value
            if name.value.startswith(b"tail."):
                assert inside == [0]
                tails += 1
            if name.value.startswith(b"head") or name.value.startswith(b"sub_mean") or name.value.startswith(b"down."):
                assert inside == [nb - 1]
        assert tails == 2 * (phase + 1)
    finally:
        L.lib().srad_drn_destroy(h)


def test_drn_loss_terms_are_what_the_reference_log_adds_up():
    """``drn_loss(..., terms=True)``: the optimised value is primary + dual_weight * dual (src/trainer.py:168-185), the logged one
    primary + dual - every call of the reference's ``Loss`` adds its value to the log, the dual terms unweighted."""
    from srad_amd.train import drn_loss
    g = torch.Generator().manual_seed(0)
    sr = [torch.rand(2, 1, 4 * 2 ** i, 4 * 2 ** i, generator=g) for i in range(3)]
    lrs = [torch.rand(2, 1, 4, 4, generator=g), torch.rand(2, 1, 8, 8, generator=g)]
    hr = torch.rand(2, 1, 16, 16, generator=g)
    sr2lr = [torch.rand(2, 1, 4, 4, generator=g), torch.rand(2, 1, 8, 8, generator=g)]
    l1 = torch.nn.functional.l1_loss
    total, logged = drn_loss(sr, lrs, hr, sr2lr, 0.1, l1, terms=True)
    primary = l1(sr[-1], hr) + l1(sr[0], lrs[0]) + l1(sr[1], lrs[1])
    dual = l1(sr2lr[0], lrs[0]) + l1(sr2lr[1], lrs[1])
    assert torch.allclose(total, primary + 0.1 * dual) and torch.allclose(logged, primary + dual)
    assert torch.equal(drn_loss(sr, lrs, hr, sr2lr, 0.1, l1), total)


def test_cosine_schedule_matches_torch():
    from srad_amd.train import cosine_lr
    p = torch.nn.Parameter(torch.zeros(1))
    opt = torch.optim.Adam([p], lr=1e-4)
    sch = torch.optim.lr_scheduler.CosineAnnealingLR(opt, 30.0, eta_min=1e-7)      # src/trainer.py:76-83
    for epoch in range(31):
        assert abs(cosine_lr(1e-4, epoch, 30.0, 1e-7) - sch.get_last_lr()[0]) < 1e-12
        opt.step()
        sch.step()


def test_adam_device_scalars_match_torch_bias_correction():
    """TensorAdam.hyper / FusedAdam.hyper: the four scalars a captured step reads from the GPU ([lr, 1 - b1^t, sqrt(1 - b2^t),
    grad_scale] of the NEXT step) are torch.optim.Adam's bias corrections (src/trainer.py:49-73 builds torch Adams)."""
    import math
    import struct
    import torch
    from srad_amd.train import TensorAdam
    p = torch.nn.Parameter(torch.zeros(4))
    opt = TensorAdam([p], lr=3e-4, betas=(0.9, 0.999))
    for step in (0, 1, 9, 999):
        opt.step_count = step
        lr, c1, c2, gs = opt.hyper(grad_scale=0.5)
        t = step + 1
        assert lr == 3e-4 and gs == 0.5
        b1, b2 = (struct.unpack("f", struct.pack("f", b))[0] for b in (0.9, 0.999))      # the betas as the C ABI passes them (fp32)
        assert abs(c1 - (1 - b1 ** t)) < 1e-12 and abs(c2 - math.sqrt(1 - b2 ** t)) < 1e-12
        assert abs(c1 - (1 - 0.9 ** t)) < 1e-5 and abs(c2 - math.sqrt(1 - 0.999 ** t)) < 1e-5


def test_launch_plan_for_gpus_n():
    """``--gpus N`` without a launcher starts N ranks (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* on 127.0.0.1); under
    torch.distributed.run (WORLD_SIZE set) or with one GPU nothing is spawned."""
    from srad_amd.launch import launch_plan
    plan = launch_plan(2, port=29555, env={})
    assert [p["RANK"] for p in plan] == ["0", "1"] and [p["LOCAL_RANK"] for p in plan] == ["0", "1"]
    assert all(p["WORLD_SIZE"] == "2" and p["MASTER_ADDR"] == "127.0.0.1" and p["MASTER_PORT"] == "29555" for p in plan)
    assert all(p["HSA_ENABLE_IPC_MODE_LEGACY"] == "0" for p in plan)
    assert launch_plan(1, env={}) == [] and launch_plan(8, env={"WORLD_SIZE": "8"}) == []
    assert len(launch_plan(8, env={})) == 8 and len({p["MASTER_PORT"] for p in launch_plan(8, env={})}) == 1


def test_bench_gpus_flag_builds_a_two_rank_launch():
    """bench.py --gpus 2 plans two child ranks (the round-1 script silently benchmarked one GPU)."""
    import importlib.util
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    args = bench.parse_args(["--gpus", "2", "--steps", "3", "--warmup", "1"])
    plan = bench.rank_plan(args, env={})
    assert len(plan) == 2 and plan[1]["RANK"] == "1" and plan[0]["WORLD_SIZE"] == "2"
    assert bench.rank_plan(bench.parse_args(["--gpus", "1"]), env={}) == []
    assert bench.rank_plan(args, env={"WORLD_SIZE": "2", "RANK": "0"}) == []          # already a rank of torch.distributed.run


def test_checkpoint_writes_what_the_evaluator_parses(tmp_path):
    """Checkpoint (src/checkpoint.py:10-28,53-58): config.txt as 'key: value' lines after a timestamp, log.txt appended
    by write_log; evaluate.infer_from_run_dir (src/evaluate.py:84-118) reads model / class / resolution / scale back."""
    from srad_amd import main as Mn
    from srad_amd.checkpoint import Checkpoint
    args = Opt.parse_train_args(["--model-type", "drn-l", "--classe", "carpet", "--resolution", "256", "--scale", "4",
                                 "--save-dir", str(tmp_path), "--batch-size", "8"])
    o = Mn.build_train_opt(args)
    assert o.test_every == 32 and o.n_colors == 3 and o.data_dir == "data/mvtec_256/carpet/train/good"
    assert o.save.startswith(str(tmp_path) + "/drn-l/mvtec_carpet_256_X4") and o.save.endswith("/")
    o.save = str(tmp_path / "renamed")
    ck = Checkpoint(o)
    ck.write_log("hello")
    ck.write_log("again", refresh=True)
    ck.done()
    lines = (tmp_path / "renamed" / "config.txt").read_text().splitlines()
    assert lines[1] == "" and "model_name: drn-l" in lines and "scale: [2, 4]" in lines and "patch_size: 256" in lines
    assert (tmp_path / "renamed" / "log.txt").read_text() == "hello\nagain\n"
    assert (tmp_path / "renamed" / "model").is_dir() and (tmp_path / "renamed" / "results").is_dir()
    inf = E.infer_from_run_dir(str(tmp_path / "renamed"))
    assert (inf["model_type"], inf["classe"], inf["resolution"], inf["scale"]) == ("drn-l", "carpet", 256, 4)
    ck2 = Checkpoint(o)                                                  # an existing run dir is appended to
    ck2.done()
    assert (tmp_path / "renamed" / "config.txt").read_text().count("model_name: drn-l") == 2


def test_bench_watchdog_prints_the_headline_line_and_ends_a_hung_rank():
    """bench.py at N > 1: a rank stuck in the training leg's all-reduce must not take the (already measured) headline line
    with it - the watchdog prints it from rank 0 with the leg marked as timed out and exits with code 3; a cancelled watchdog
    does nothing."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    prog = ("import sys, time, json; sys.path.insert(0, %r); import bench\n"
            "r = {'metric': 'm', 'value': 1.0}\n"
            "g = bench.arm_train_watchdog(r, 0, 30.0); g.cancel()\n"
            "bench.arm_train_watchdog(r, int(sys.argv[1]), 0.3, linger_s=0.2)\n"
            "time.sleep(20)\nprint('not reached')\n") % root
    for rank in (0, 1):
        out = subprocess.run([sys.executable, "-c", prog, str(rank)], capture_output=True, text=True, timeout=60)
        assert out.returncode == 3, (out.returncode, out.stderr[-500:])
        assert "not reached" not in out.stdout
        if rank == 0:
            line = json.loads(out.stdout.strip().splitlines()[-1])
            assert line["value"] == 1.0 and "timeout" in line["train"]["error"]
        else:
            assert out.stdout.strip() == ""
