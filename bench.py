#!/usr/bin/env python3
"""bench.py - headline benchmark: DRCT-L x4 super-resolution forward, HR Mpixels/s.

Workload (BASELINE.json configs[1], "C2"): DRCT-L on a 128 px HR grid tile, scale x4, batch 4 per
GPU, bf16 MFMA with fp32 accumulate -> LR input [4,1,32,32] fp32 in [0,255], output [4,1,128,128].
One "step" = one forward of that batch through the HIP engine, inputs already resident in HBM.
Synthetic data; reference-style random init of the full 12-RDG architecture (27.38 M params).

  python bench.py [--gpus N] [--steps K] [--warmup W] [--dtype bf16|fp32]
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

N > 1: one process per GPU, every rank runs its own batch (image-parallel, no data-path
collective; SURVEY.md §8(e) eval row) -> weak scaling; barrier + synchronize on both sides of the
timed region, MAX over ranks.  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK = {"bf16": 2.5e15, "fp32": 157.3e12, "bf16x3": 2.5e15}      # dense MFMA peaks, MI355X_MICROARCH.md (split-bf16 runs on the bf16 pipe)
HBM_PEAK = 8.0e12
CSRC = os.path.join(ROOT, "anomaly-detection-super-resolution_amd", "csrc")


def kernel_source_sha() -> str:
    """sha256 over the HIP sources: a committed PMC profile is only quoted while it describes THESE kernels."""
    import hashlib
    h = hashlib.sha256()
    for name in sorted(os.listdir(CSRC)):
        if name.endswith((".hip", ".h")):
            h.update(name.encode())
            h.update(open(os.path.join(CSRC, name), "rb").read())
    return h.hexdigest()[:16]


def pmc_profile(kernel: str, leg: str = "c2"):
    """(counters of `kernel`, file, stale?) from the newest profiles/r*_<leg>_pmc_counters.json (tools/pmc_summary.py: separate
    rocprofv3 --pmc passes of `bench.py --only <leg>`, tools/pmc_passes.sh)."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", f"r*_{leg}_pmc_counters.json")))
    if not files and leg == "c2":
        files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_counters.json")))
    if not files:
        return None, None, None
    try:
        pmc = json.load(open(files[-1]))
        return pmc["kernels"].get(kernel), os.path.relpath(files[-1], ROOT), pmc.get("kernel_source_sha") != kernel_source_sha()
    except Exception:
        return None, None, None


def kernel_roofline(prof: dict, reps: int, leg: str, peak: float, dtype_note: str = ""):
    """`roofline` object of a leg from its per-class HIP-event profile (L.prof_collect): the class with the most device time,
    its algorithmic FLOPs (2 M N K per launch, summed by the launchers) over its summed launch durations, against the dense MFMA
    peak of the leg's dtype; counters from the leg's own PMC passes when their source hash still matches."""
    if not prof:
        return None
    total_ms = sum(v["ms"] for v in prof.values())
    dom = max(prof, key=lambda k: prof[k]["ms"])
    d = prof[dom]
    ach = d["flops"] / (d["ms"] * 1e-3)
    roof = {"kernel": dom, "bound": "mfma", "achieved": round(ach / 1e12, 2), "peak": peak / 1e12, "unit": "TFLOP/s",
            "frac": round(ach / peak, 4), "traffic": None, "launches_per_step": d["launches"] // reps,
            "avg_launch_us": round(d["ms"] * 1e3 / d["launches"], 3),
            "algorithmic_gflop_per_launch": round(d["flops"] / d["launches"] / 1e9, 4),
            "share_of_kernel_time": round(d["ms"] / total_ms, 3)}
    pmc, pmc_file, stale = pmc_profile(dom, leg)
    if pmc is not None:
        roof["pmc_source"], roof["pmc_stale"] = pmc_file, bool(stale)
        if not stale:
            roof["traffic"] = pmc.get("hbm_bytes_per_launch")
            for k in ("mfma_busy", "valu_busy", "wait_frac", "lds_bank_conflict_frac", "dispatch_us"):
                if k in pmc:
                    roof[k] = pmc[k]
    return roof


def profile_eager(fn, reps: int):
    """HIP-event class profile of `reps` eager calls of fn (one untimed call first)."""
    import torch
    from srad_amd import _lib as L
    L.prof_enable(True)
    fn()
    torch.cuda.synchronize()
    L.prof_collect()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    prof = L.prof_collect()
    L.prof_enable(False)
    return prof


def kernel_table(prof: dict, reps: int):
    return {k: {"launches_per_step": v["launches"] // reps, "avg_us": round(v["ms"] * 1e3 / v["launches"], 2), "ms_per_step": round(v["ms"] / reps, 3),
                "tflops": round(v["flops"] / (v["ms"] * 1e-3) / 1e12, 2)} for k, v in sorted(prof.items(), key=lambda kv: -kv[1]["ms"])}


def usable_cores() -> int:
    """CPU cores this process may actually use (cgroup quota / affinity), not the host's total."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return max(1, n)


class Opt:
    n_colors, img_size, window_size, upscale = 1, 32, 8, 4
    embed_dim, depths, num_heads, mlp_ratio, img_range = 180, (6,) * 12, (6,) * 12, 2, 1.0
    upsampler, resi_connection = "pixelshuffle", "1conv"
    precision, use_graph = "bf16", True


def anomaly_eval_leg(model, args, torch):
    """Second metric of BASELINE.json: anomaly-eval images/s = image pairs fully scored per second
    (SR forward + truncating u8 + SSIM window sweep + MSE + PSNR + the three AUCs) on an MVTec-grid sized
    synthetic test split (21 good + 57 bad pairs, 128 px HR), and the AUCs against the CPU oracle."""
    import numpy as np
    from srad_amd import evaluate as E
    from srad_amd.spec import synth_pairs

    class EvalOpt:
        rgb_range = 255.0
    y, sr_u8, hr_u8 = synth_pairs(21, 57, 128, 1, seed=0)
    pairs = []
    for s_img, h_img in zip(sr_u8, hr_u8):           # LR = 4x4 box average of the (defective) image
        lr = s_img.reshape(32, 4, 32, 4, 1).astype(np.float32).mean((1, 3))
        pairs.append((np.clip(np.rint(lr), 0, 255).astype(np.uint8), h_img))
    good, bad = pairs[:21], pairs[21:]
    was_graph = model.use_graph
    model.use_graph = False                          # batches of 8 images: not the captured shape
    import contextlib, io
    with torch.no_grad(), contextlib.redirect_stdout(io.StringIO()):
        E.evaluate_on_test(EvalOpt, model, good, bad)                     # warm-up (allocations)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        reps = 3
        for _ in range(reps):
            got = E.evaluate_on_test(EvalOpt, model, good, bad)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / reps
    model.use_graph = was_graph
    out = {"images_per_s": round(len(pairs) / dt, 1), "n_images": len(pairs), "hr_px": 128,
           "window_sizes": len(got["window_sizes"]), "ms_per_split": round(dt * 1e3, 2),
           "auc": {k: round(got[k], 4) for k in ("auc_ssim", "auc_mse", "auc_psnr")}, "best_ws": got["best_ws"]}
    # CPU baseline of the scorer: the reference's per-pixel-loop ssim_numpy semantics (oracle, literal
    # loop) on a bounded sample: 1 pair x all window sizes, single core
    from oracle import scorer_ref as O          # cpu_baseline leg only: the checker, never the thing measured
    t0 = time.perf_counter()
    sizes = O.sweep_window_sizes(128)
    for ws in sizes:
        O.ssim_numpy(hr_u8[0].astype(np.float32) / 255.0, sr_u8[0].astype(np.float32) / 255.0, ws, fast=False)
    cpu_pair = time.perf_counter() - t0
    out["cpu_scorer_baseline"] = {"pairs_per_s": round(1.0 / cpu_pair, 4), "cores": 1, "kind": "port",
                                  "sample": f"1 pair x {len(sizes)} window sizes, literal per-pixel loop, {cpu_pair:.1f} s"}
    if not args.no_cpu_baseline:
        # AUC parity: the same split through the CPU oracle (fp32 forward, truncating u8, window sweep)
        from oracle import sr_ref as R
        torch.set_num_threads(usable_cores())
        sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
        o_sr = []
        with torch.no_grad():
            for i in range(0, len(pairs), 6):
                x = torch.from_numpy(np.stack([p[0] for p in pairs[i:i + 6]])).permute(0, 3, 1, 2).float()
                o_sr += [np.transpose(O.to_u8_trunc(t), (1, 2, 0)) for t in R.drct_forward(sd, x, model.cfg).numpy()]
        ref = O.evaluate_pairs(y, o_sr, hr_u8)
        out["auc_oracle"] = {k: round(ref[k], 4) for k in ("auc_ssim", "auc_mse", "auc_psnr")}
        out["auc_abs_diff"] = round(max(abs(got[k] - ref[k]) for k in ("auc_ssim", "auc_mse", "auc_psnr")), 5)
        # the modes that own the AUC +-0.002 claim on the same split: split-bf16 (the evaluator's default) and exact fp32
        from srad_amd.nets import DRCT
        for mode in ("bf16x3", "fp32"):
            o32 = Opt()
            o32.precision, o32.use_graph = mode, False
            m32 = DRCT(o32).to(next(model.parameters()).device).eval()
            m32.load_state_dict(model.state_dict())
            with torch.no_grad(), contextlib.redirect_stdout(io.StringIO()):
                g32 = E.evaluate_on_test(EvalOpt, m32, good, bad)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(reps):
                    E.evaluate_on_test(EvalOpt, m32, good, bad)
                torch.cuda.synchronize()
                dt32 = (time.perf_counter() - t0) / reps
            out[f"auc_abs_diff_{mode}_mode"] = round(max(abs(g32[k] - ref[k]) for k in ("auc_ssim", "auc_mse", "auc_psnr")), 5)
            out[f"images_per_s_{mode}_mode"] = round(len(pairs) / dt32, 1)
            del m32
        out["auc_parity_note"] = (f"HIP engine vs fp32 CPU oracle, same random-init weights (AUC is near chance, so rankings are "
                                  f"noise-sensitive).  The evaluator's DEFAULT is the split-bf16 mode (evaluate --dtype bf16x3): "
                                  f"bar +-0.002; '{args.dtype}' is the opt-in fast mode this leg's headline images/s times")
    return out


def train_leg(args, torch, dist, dev, world, rank):
    """BASELINE config C4: DRCT-L x4 training step (forward + L1 + backward + Adam), 128 px HR, 8 images per GPU
    (global batch 8 x N), data-parallel with the per-RDG gradient buckets all-reduced over RCCL while the backward
    runs.  Runs on every rank; returns the rank-0 summary."""
    from srad_amd import _lib as L
    from srad_amd.nets import DRCT
    from srad_amd.train import FusedAdam, GradReducer, train_step
    o = Opt()
    o.precision, o.use_graph = args.dtype, False
    torch.manual_seed(1)                                   # identical replicas ...
    m = DRCT(o).to(dev).train()
    m.enable_training()
    torch.manual_seed(1 + rank)                            # ... but per-rank DropPath draws (seed + rank)
    opt = FusedAdam(m, lr=1e-4)
    red = GradReducer().attach(m) if world > 1 else None
    B = args.train_batch
    g = torch.Generator(device="cpu").manual_seed(100 + rank)
    lr_img = (torch.rand(B, 1, 32, 32, generator=g) * 255.0).to(dev)
    hr_img = (torch.rand(B, 1, 128, 128, generator=g) * 255.0).to(dev)
    steps, warm = args.train_steps, 3
    graphed = world == 1 and not args.no_train_graph
    if graphed:       # one GPU: the whole step is one replayed hipGraph (two eager steps, then the capture)
        from srad_amd.train import GraphedTrainStep
        gstep = GraphedTrainStep(m, opt, warmup=2)
        step_fn = lambda: gstep(lr_img, hr_img)
    else:             # data parallel: the bucket hooks launch RCCL all-reduces from the host during the backward
        step_fn = lambda: train_step(m, lr_img, hr_img, opt, red)
    losses = []
    for _ in range(warm):
        losses.append(step_fn())
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    host_first = 0.0
    for i in range(steps):
        losses.append(step_fn())
        if i == 0:
            host_first = time.perf_counter() - t0   # enqueue time of one step into an empty queue
    host_el = time.perf_counter() - t0            # the host has enqueued every step (it never waits for the GPU)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    el = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([el], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        el = float(t.item())
    out = {"workload": f"C4: DRCT-L x4 train step, 128px HR, {B} images per GPU (global batch {B * world}), L1 + Adam, DropPath 0.1",
           "ms_per_step": round(el / steps * 1e3, 3), "steps": steps,
           "launch": "one hipGraph replay per step" if graphed else "eager (two streams)",
           "host_enqueue_ms_per_step": round(host_el / steps * 1e3, 3), "host_enqueue_ms_first_step": round(host_first * 1e3, 3),
           "hr_mpixels_per_s": round(world * B * 128 * 128 * steps / el / 1e6, 3),
           "images_per_s": round(world * B * steps / el, 2),
           "loss_first_last": [round(float(losses[0]), 4), round(float(losses[-1]), 4)],
           "grad_allreduce": ({"backend": dist.get_backend(), "collective": "all-reduce(SUM) per RDG bucket on a side stream, "
                               "overlapped with the backward; 1/world folded into Adam", "buckets": len([b for b in m.grad_buckets if b[1] > 0]),
                               "bytes_per_step": int(4 * sum(n for _, n in m.grad_buckets)),
                               "largest_bucket_bytes": int(4 * max(n for _, n in m.grad_buckets))}
                              if world > 1 else "none (1 GPU)"),
           "droppath_seed": f"1 + rank"}
    if rank == 0:
        L.prof_enable(True)
        train_step(m, lr_img, hr_img, opt, red)
        torch.cuda.synchronize()
        L.prof_collect()
        reps = 3
        for _ in range(reps):
            train_step(m, lr_img, hr_img, opt, red)
        torch.cuda.synchronize()
        prof = L.prof_collect()
        L.prof_enable(False)
        out["kernels"] = {k: {"launches_per_step": v["launches"] // reps, "avg_us": round(v["ms"] * 1e3 / v["launches"], 2),
                              "ms_per_step": round(v["ms"] / reps, 3),
                              "tflops": round(v["flops"] / (v["ms"] * 1e-3) / 1e12, 2)}
                          for k, v in sorted(prof.items(), key=lambda kv: -kv[1]["ms"])}
        fl = 3.0 * m.flops(B, 32, 32)
        out["algorithmic_gflop_per_step"] = round(fl / 1e9, 1)
        out["model_tflops"] = round(fl * world / (el / steps) / 1e12, 2)
        out["roofline"] = kernel_roofline(prof, reps, "c4", PEAK[args.dtype])
    del m, opt
    torch.cuda.empty_cache()
    return out


def scorer_leg(torch, dev):
    """Roofline of the reconstruction-error scorer (HBM-bound, algorithmic bytes = two fp32 luminance
    planes per pair per window size = 8 H W B).  Two shapes: the MVTec-grid split of the anomaly-eval leg (78 pairs, 128 px,
    13 window sizes) and one C5 tile pair (1024 px, 102 window sizes); u8 pairs already resident in HBM, timed with
    events on the launch stream, all window sizes in one call (one SAT build + one evaluation launch per window size)."""
    from srad_amd import metrics as M
    out = {}
    pmc_cases, pmc_file, stale = {}, None, None
    try:
        import glob
        files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_scorer_pmc.json")))
        if files:
            pj = json.load(open(files[-1]))
            pmc_cases, pmc_file, stale = pj.get("cases", {}), os.path.relpath(files[-1], ROOT), pj.get("kernel_source_sha") != kernel_source_sha()
    except Exception:
        pmc_cases = {}
    for tag, n, px in (("grid_128px", 78, 128), ("tile_1024px", 2, 1024)):
        g = torch.Generator(device="cpu").manual_seed(5)
        hr = torch.randint(0, 256, (n, px, px, 1), generator=g, dtype=torch.uint8).to(dev)
        sr = (hr.int() + torch.randint(-6, 7, hr.shape, generator=g).to(dev)).clamp(0, 255).to(torch.uint8)
        sizes = M.sweep_window_sizes(px)
        for _ in range(3):                                     # (one call left the 0.3 ms grid case at the mercy of the leg before it:
            M.score_pairs(sr, hr, sizes)                       #  0.57 ms inside a full run against 0.29 alone)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 20 if px <= 128 else 5
        e0.record()
        for _ in range(reps):
            M.score_pairs(sr, hr, sizes)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        algo = 8.0 * px * px * n * len(sizes)
        roof = {"bound": "hbm", "achieved": round(algo / (ms * 1e-3) / 1e9, 1), "peak": HBM_PEAK / 1e9, "unit": "GB/s",
                "frac": round(algo / (ms * 1e-3) / HBM_PEAK, 4), "traffic": None, "algorithmic_bytes": int(algo)}
        pmc = pmc_cases.get(tag)
        if pmc is not None:
            # FETCH_SIZE + WRITE_SIZE of one score_pairs call (tools/pmc_scorer.sh: separate rocprofv3 --pmc passes; the 8-byte-per-lane
            # read factor is calibrated in the same pass); quoted only while the profile's source hash equals these kernels
            roof["pmc_source"], roof["pmc_stale"] = pmc_file, bool(stale)
            if not stale and pmc.get("traffic_bytes") is not None:
                roof["traffic"] = pmc["traffic_bytes"]
                roof["traffic_over_algorithmic"] = pmc.get("traffic_over_algorithmic")
        out[tag] = {"pairs": n, "hr_px": px, "window_sizes": len(sizes), "ms": round(ms, 3),
                    "pairs_x_windows_per_s": round(n * len(sizes) / (ms * 1e-3), 1), "roofline": roof}
    return out


def c5_leg(args, torch, dev):
    """BASELINE config C5: DRCT-L eval on one 1024 px HR tile (LR [1,1,256,256], window 64 -> 4096-token windows)."""
    from srad_amd import _lib as L
    from srad_amd.nets import DRCT
    o = Opt()
    o.img_size, o.window_size, o.precision, o.use_graph = 256, 64, args.dtype, True
    torch.manual_seed(1)
    m = DRCT(o).to(dev).eval()
    x = torch.rand(1, 1, 256, 256, device=dev) * 255.0
    steps = 5
    with torch.no_grad():
        for _ in range(3):
            m(x)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            m(x)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / steps
        m.use_graph = False
        L.prof_enable(True)
        m(x)
        torch.cuda.synchronize()
        L.prof_collect()
        m(x)
        torch.cuda.synchronize()
        prof = L.prof_collect()
        L.prof_enable(False)
    fl = m.flops(1, 256, 256)
    out = {"workload": "C5: DRCT-L x4 forward, one 1024 px HR tile (LR [1,1,256,256]), window 64", "ms_per_tile": round(dt * 1e3, 2),
           "hr_mpixels_per_s": round(1024 * 1024 / dt / 1e6, 2), "algorithmic_gflop": round(fl / 1e9, 1),
           "model_tflops": round(fl / dt / 1e12, 1),
           "kernels": {k: {"launches": v["launches"], "ms": round(v["ms"], 3), "tflops": round(v["flops"] / (v["ms"] * 1e-3) / 1e12, 1)}
                       for k, v in sorted(prof.items(), key=lambda kv: -kv[1]["ms"])},
           "roofline": kernel_roofline(prof, 1, "c5", PEAK[args.dtype])}
    if args.dtype == "bf16":
        # the same tile in the parity-grade mode (split-bf16: split instances of ln_qkv, the window attention and mlp_block), and how
        # far the bf16 output is from it
        with torch.no_grad():
            y16 = m(x)
            o.precision = "bf16x3"
            m3 = DRCT(o).to(dev).eval()
            m3.load_state_dict(m.state_dict())
            y3 = m3(x)
            for _ in range(2):
                m3(x)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(3):
                m3(x)
            torch.cuda.synchronize()
            dt3 = (time.perf_counter() - t0) / 3
        rng = float(y3.max() - y3.min())
        out["parity_mode"] = {"dtype": "bf16x3", "ms_per_tile": round(dt3 * 1e3, 2), "hr_mpixels_per_s": round(1024 * 1024 / dt3 / 1e6, 2),
                              "bf16_max_err_over_range_vs_this_mode": round(float((y16 - y3).abs().max()) / rng, 5)}
        del m3
    del m
    torch.cuda.empty_cache()
    return out


def c3_leg(args, torch, dev):
    """BASELINE config C3: DRN-L x4 forward, carpet-shaped RGB input, 256 px HR, batch 8 (LR [8,3,64,64])."""
    from srad_amd.nets import DRN
    from srad_amd import _lib as L      # noqa: F401  (profile_eager)

    class DrnOpt:
        n_colors, n_blocks, n_feats, negval, rgb_range, scale = 3, 40, 20, 0.2, 255.0, [2, 4]
        precision, use_graph = args.dtype, True
    torch.manual_seed(1)
    m = DRN(DrnOpt()).to(dev).eval()
    x = torch.rand(8, 3, 64, 64, device=dev) * 255.0
    steps = 20
    with torch.no_grad():
        for _ in range(3):
            m(x)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            m(x)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / steps
        m.use_graph = False
        prof = profile_eager(lambda: m(x), 5)
    fl = m.flops(8, 64, 64)
    out = {"workload": "C3: DRN-L x4 forward, RGB, 256 px HR, batch 8 (LR [8,3,64,64])", "ms_per_batch": round(dt * 1e3, 3),
           "hr_mpixels_per_s": round(8 * 256 * 256 / dt / 1e6, 2), "algorithmic_gflop": round(fl / 1e9, 1),
           "model_tflops": round(fl / dt / 1e12, 1), "kernels": kernel_table(prof, 5),
           "roofline": kernel_roofline(prof, 5, "c3", PEAK[args.dtype])}
    del m
    torch.cuda.empty_cache()
    return out


def c3_cpu_baseline(args, torch, dev, c3):
    """the reference's --device cpu path beside every config - the oracle's DRN forward (stock torch CPU kernels,
    fp32) on a bounded sample: ONE image of the C3 batch (the batch of 8 would take ~1 s per forward).  Runs after every GPU leg:
    a CPU thread pool left spinning slows the host-side launch loops of the eager legs (seen: DRN training 22 -> 26 ms)."""
    from oracle import sr_ref as R
    from srad_amd.nets import DRN

    class DrnOpt:
        n_colors, n_blocks, n_feats, negval, rgb_range, scale = 3, 40, 20, 0.2, 255.0, [2, 4]
        precision, use_graph = args.dtype, False
    torch.manual_seed(1)
    m = DRN(DrnOpt()).to(dev).eval()
    x = torch.rand(8, 3, 64, 64, device=dev) * 255.0
    torch.set_num_threads(usable_cores())
    sd = {k: v.detach().cpu() for k, v in m.state_dict().items()}
    xc = x[:1].cpu()
    with torch.no_grad():
        t0 = time.perf_counter()
        ref = R.drn_forward(sd, xc, m.cfg)
        first = time.perf_counter() - t0
        n = max(1, min(10, int(args.cpu_seconds / max(first, 1e-3))))
        t0 = time.perf_counter()
        for _ in range(n):
            R.drn_forward(sd, xc, m.cfg)
        cpu_t = (time.perf_counter() - t0) / n
        got = m(x[:1])[-1].cpu()
    c3["cpu_baseline"] = {"value": round(256 * 256 / cpu_t / 1e6, 4), "unit": "HR Mpixels/s", "cores": torch.get_num_threads(), "kind": "port",
                          "sample": f"{n} forwards of ONE image of the C3 shape (fp32, torch CPU kernels via oracle/sr_ref.py), {cpu_t * 1e3:.0f} ms each"}
    c3["speedup_vs_cpu"] = round(c3["hr_mpixels_per_s"] / c3["cpu_baseline"]["value"], 1)
    c3["max_rel_err_vs_cpu_fp32"] = float(f"{float((got - ref[-1]).abs().max() / ref[-1].abs().max()):.3e}")
    del m
    torch.cuda.empty_cache()


def drn_train_leg(args, torch, dev):
    """DRN-L x4 training step at the C3 shape (src/trainer.py:161-205 with dual_model=True): SR net + two dual regression
    models, composite loss, one fused Adam for the SR net and a torch Adam per dual model; batch 8, RGB, 256 px HR."""
    from srad_amd.nets import DRN, DownBlock
    from srad_amd.train import FusedAdam, GraphedDrnTrainStep, TensorAdam, drn_train_step

    class DrnOpt:
        n_colors, n_blocks, n_feats, negval, rgb_range, scale = 3, 40, 20, 0.2, 255.0, [2, 4]
        precision, use_graph = args.dtype, False
    torch.manual_seed(1)
    m = DRN(DrnOpt()).to(dev).train()
    m.enable_training()
    duals = [DownBlock(DrnOpt()).to(dev) for _ in DrnOpt.scale]
    opt = FusedAdam(m, lr=1e-4, weight_decay=1e-8)
    dopts = [TensorAdam(d.parameters(), lr=1e-4, weight_decay=1e-8) for d in duals]      # the engine's Adam kernel, as the Trainer uses it
    B = 8
    lrs = [torch.rand(B, 3, 64, 64, device=dev) * 255, torch.rand(B, 3, 128, 128, device=dev) * 255]
    hr = torch.rand(B, 3, 256, 256, device=dev) * 255
    for _ in range(2):
        drn_train_step(m, duals, lrs, hr, opt, dopts)
    torch.cuda.synchronize()
    steps = 5
    t0 = time.perf_counter()
    for _ in range(steps):
        loss = drn_train_step(m, duals, lrs, hr, opt, dopts)
    torch.cuda.synchronize()
    dt_eager = (time.perf_counter() - t0) / steps
    # the step as the Trainer runs it on one GPU with '1*L1': one hipGraph per step (train.GraphedDrnTrainStep).  Counter
    # passes set SRAD_BENCH_NO_STEP_GRAPH: rocprofv3 --pmc segfaulted (host side, deep recursion) instantiating this ~2000-node
    # graph, and the per-kernel counters are those of the eager launches anyway.
    graphed = not os.environ.get("SRAD_BENCH_NO_STEP_GRAPH")
    dt = dt_eager
    if graphed:
        gstep = GraphedDrnTrainStep(m, duals, opt, dopts, warmup=1)
        for _ in range(3):
            gstep(lrs, hr)
        torch.cuda.synchronize()
        steps = 10
        t0 = time.perf_counter()
        for _ in range(steps):
            loss = gstep(lrs, hr)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / steps
    fl = 3.0 * m.flops(B, 64, 64)
    prof = profile_eager(lambda: drn_train_step(m, duals, lrs, hr, opt, dopts), 3)
    out = {"workload": "DRN-L x4 train step (SR net + 2 dual models, composite loss, Adam), RGB, 256 px HR, batch 8",
           "ms_per_step": round(dt * 1e3, 2), "launch": "one hipGraph per step (as the Trainer on one GPU)" if graphed else "eager", "eager_ms_per_step": round(dt_eager * 1e3, 2),
           "images_per_s": round(B / dt, 1), "hr_mpixels_per_s": round(B * 256 * 256 / dt / 1e6, 2),
           "model_tflops": round(fl / dt / 1e12, 1), "loss": round(float(loss), 4), "kernels": kernel_table(prof, 3),
           "roofline": kernel_roofline(prof, 3, "drn_train", PEAK[args.dtype])}
    del m, duals, opt, dopts
    torch.cuda.empty_cache()
    return out


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp32", "bf16x3"])
    ap.add_argument("--batch", type=int, default=4)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
    ap.add_argument("--no-eval", action="store_true", help="skip the anomaly-eval images/s leg")
    ap.add_argument("--no-train", action="store_true", help="skip the C4 training-step leg")
    ap.add_argument("--no-train-graph", action="store_true", help="C4 leg: eager launches instead of one hipGraph per step")
    ap.add_argument("--train-batch", type=int, default=8)
    ap.add_argument("--train-steps", type=int, default=10)
    ap.add_argument("--only", default="", choices=["", "c3", "c4", "c5", "drn_train", "scorer"],
                    help="run ONE secondary leg and print its object (the program of the per-leg counter passes, tools/pmc_passes.sh)")
    return ap.parse_args(argv)


def rank_plan(args, env=None):
    """Environments of the ranks this process must start: ``--gpus N`` with no launcher around it -> N entries; a rank of
    ``torch.distributed.run`` (WORLD_SIZE set) or N = 1 -> [] (run in this process)."""
    from srad_amd.launch import launch_plan
    return launch_plan(args.gpus, env=env)


PARITY_MODES = {
    # name: (what, peak the roofline is priced against, MFMA issues per algorithmic product)
    "bf16x3": ("C2 forward in the split-bf16 parity mode (hi + lo bf16 operands, three bf16 MFMAs per product, fp32 accumulate; "
               "the same two fused launches per Swin block as the bf16 headline)", PEAK["bf16"]),
    "fp32": ("C2 forward in the fp32 mode (exact-fp32 MFMA, six launches per Swin block)", PEAK["fp32"]),
}


def parity_mode_leg(model, x, y_bf16, args, torch, dev, mode="bf16x3"):
    """A mode that meets the 1e-3 parity bar on the same C2 forward, timed like the headline (graph replay), with its own
    roofline: "bf16x3" = split-bf16 (the evaluator's default; algorithmic FLOPs against the bf16 dense peak - the three MFMAs per
    product are the mode's cost, not extra work), "fp32" = exact-fp32 MFMA v_mfma_f32_16x16x4_f32 against the 157.3 TFLOP/s peak."""
    from srad_amd import _lib as L
    from srad_amd.nets import DRCT
    what, peak = PARITY_MODES[mode]
    o32 = Opt()
    o32.precision, o32.use_graph = mode, not args.no_graph
    m32 = DRCT(o32).to(dev).eval()
    m32.load_state_dict(model.state_dict())
    steps = max(10, min(args.steps, 50))
    with torch.no_grad():
        for _ in range(3):
            y32 = m32(x)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            y32 = m32(x)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / steps
        m32.use_graph = False
        L.prof_enable(True)
        m32(x)
        torch.cuda.synchronize()
        L.prof_collect()
        reps = 5
        for _ in range(reps):
            m32(x)
        torch.cuda.synchronize()
        prof = L.prof_collect()
        L.prof_enable(False)
    B, _, H, W = x.shape
    flops = m32.flops(B, H, W)
    dom = max(prof, key=lambda k: prof[k]["ms"])
    d = prof[dom]
    ach = d["flops"] / (d["ms"] * 1e-3)
    out = {"dtype": mode, "what": what,
           "ms_per_step": round(dt * 1e3, 4), "hr_mpixels_per_s": round(B * H * 4 * W * 4 / dt / 1e6, 3),
           "model_tflops": round(flops / dt / 1e12, 2),
           "roofline": {"kernel": dom, "bound": "mfma", "achieved": round(ach / 1e12, 2), "peak": peak / 1e12,
                        "unit": "TFLOP/s", "frac": round(ach / peak, 4), "traffic": None,
                        "launches_per_step": d["launches"] // reps, "avg_launch_us": round(d["ms"] * 1e3 / d["launches"], 3)},
           "headline_bf16_vs_this_mode_max_rel": float(f"{float((y_bf16 - y32).abs().max() / y32.abs().max()):.3e}")}
    return out, y32, m32


def run_only(args):
    """One secondary leg on one GPU, eager where that matters to a profiler (counter passes: tools/pmc_passes.sh)."""
    import torch
    import torch.distributed as dist
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    args.no_cpu_baseline = True
    if args.only == "c3":
        out = c3_leg(args, torch, dev)
    elif args.only == "c5":
        out = c5_leg(args, torch, dev)
    elif args.only == "c4":
        args.no_train_graph, args.train_steps = True, 3
        out = train_leg(args, torch, dist, dev, 1, 0)
    elif args.only == "drn_train":
        out = drn_train_leg(args, torch, dev)
    else:
        out = scorer_leg(torch, dev)
    print(json.dumps({args.only: out}), flush=True)


def arm_train_watchdog(result, rank, limit_s, linger_s=5.0):
    """The data-parallel training leg is the one leg with collectives on its data path: if a rank dies or hangs in it, the
    others wait in an all-reduce forever and the headline line (already measured, in `result`) would never be printed.  This
    timer prints it from rank 0 with the leg marked as timed out and ends the process (exit code 3); the other ranks leave
    `linger_s` later, so that the launcher does not end rank 0 before it has printed.  cancel() the returned timer when the
    leg is through."""
    import threading

    def give_up():
        if rank == 0:
            result["train"] = {"error": f"timeout: the data-parallel training leg did not finish within {limit_s:.0f} s "
                                        "(a rank hung or died in it); line printed by the watchdog"}
            print(json.dumps(result), flush=True)
        else:
            time.sleep(linger_s)
        os._exit(3)
    guard = threading.Timer(limit_s, give_up)
    guard.daemon = True
    guard.start()
    return guard


def run_rank(args):
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        print(f"warning: --gpus {args.gpus} but WORLD_SIZE {world}; the launcher's WORLD_SIZE wins", file=sys.stderr)
    n_gpus = world
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the HIP engine has no CPU fallback)")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)       # "nccl" is RCCL on ROCm (xGMI between the GPUs of a node)

    from srad_amd import _lib as L
    from srad_amd.nets import DRCT

    opt = Opt()
    opt.precision = args.dtype
    opt.use_graph = not args.no_graph
    torch.manual_seed(1)                                   # reference seed (src/main.py:41,89): identical replicas
    model = DRCT(opt).to(dev).eval()
    B, H, W, s = args.batch, 32, 32, 4
    g = torch.Generator(device="cpu").manual_seed(1 + rank)
    x = (torch.rand(B, 1, H, W, generator=g) * 255.0).to(dev)

    def barrier():
        if world > 1:
            dist.barrier()

    with torch.no_grad():
        for _ in range(max(args.warmup, 3)):               # >= 3: eager, graph capture, first replay
            y = model(x)
        torch.cuda.synchronize()
        barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            y = model(x)
        torch.cuda.synchronize()
        barrier()
        t1 = time.perf_counter()
    elapsed = t1 - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    hr_px = B * H * s * W * s
    value = n_gpus * hr_px * args.steps / elapsed / 1e6

    result = {
        "metric": "HR Mpixels/sec DRCT-L x4 @128px HR (SR forward)",
        "value": round(value, 3),
        "unit": "HR Mpixels/s",
        "n_gpus": n_gpus,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(elapsed / args.steps * 1e3, 4),
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": args.dtype,
        "data": "synthetic",
        "config": {"workload": "C2: DRCT-L x4 forward, MVTec-grid shape, 128px HR, batch 4 per GPU "
                               "(LR [4,1,32,32] fp32 -> HR [4,1,128,128]), window 8, 12 RDG x 5 Swin blocks",
                   "per_gpu_batch": B, "parallelism": f"image-parallel x{n_gpus} (no collective)",
                   "hipgraph": bool(opt.use_graph),
                   "ranks": f"{world} process(es), one per GPU" + (f", backend {dist.get_backend()} (RCCL)" if world > 1 else "")},
    }

    if not args.no_train:
        guard = None
        if world > 1:
            guard = arm_train_watchdog(result, rank, float(os.environ.get("SRAD_BENCH_TRAIN_LIMIT_S", "420")))
        try:
            torch.manual_seed(1)                           # the replica's weights: same seed on every rank ...
            result["train"] = train_leg(args, torch, dist, dev, world, rank)
        except Exception as e:          # the headline line must survive a failure of this extra leg
            result["train"] = {"error": f"{type(e).__name__}: {e}"}
        if guard is not None:
            guard.cancel()

    if rank == 0:
        flops = model.flops(B, H, W)
        result["algorithmic_gflop_per_step"] = round(flops / 1e9, 2)
        result["model_tflops"] = round(flops * args.steps / (elapsed) / 1e12, 2)
        # ---- per-kernel timing with HIP events on the launch stream (eager, same workload) ----
        model.use_graph = False
        L.prof_enable(True)
        reps = 10
        with torch.no_grad():
            model(x)
            torch.cuda.synchronize()
            L.prof_collect()
            for _ in range(reps):
                model(x)
        torch.cuda.synchronize()
        prof = L.prof_collect()
        L.prof_enable(False)
        model.use_graph = opt.use_graph
        total_ms = sum(v["ms"] for v in prof.values())
        dom = max(prof, key=lambda k: prof[k]["ms"])
        d = prof[dom]
        achieved = d["flops"] / (d["ms"] * 1e-3)
        pmc, pmc_file, stale = pmc_profile(dom)
        roof = {
            "kernel": dom, "bound": "mfma", "achieved": round(achieved / 1e12, 2), "peak": PEAK[args.dtype] / 1e12,
            "unit": "TFLOP/s", "frac": round(achieved / PEAK[args.dtype], 4),
            "traffic": None if (pmc is None or stale) else pmc.get("hbm_bytes_per_launch"),
            "launches_per_step": d["launches"] // reps,
            "avg_launch_us": round(d["ms"] * 1e3 / d["launches"], 3),
            "algorithmic_gflop_per_launch": round(d["flops"] / d["launches"] / 1e9, 4),
            "algorithmic_mbytes_per_launch": round(d["bytes"] / d["launches"] / 1e6, 3),
            "share_of_kernel_time": round(d["ms"] / total_ms, 3),
        }
        if pmc is not None:
            # counters come from separate rocprofv3 --pmc passes of this same workload (a PMC pass cannot run inside this
            # process); they are quoted only while the profile's source hash equals the kernels that just ran
            roof["pmc_source"] = pmc_file
            roof["pmc_stale"] = bool(stale)
            if not stale:
                for k in ("mfma_busy", "mfma_busy_note", "valu_busy", "wait_frac", "lds_bank_conflict_frac", "dispatch_us"):
                    if k in pmc:
                        roof[k] = pmc[k]
        result["roofline"] = roof
        result["kernels"] = {k: {"launches_per_step": v["launches"] // reps,
                                 "avg_us": round(v["ms"] * 1e3 / v["launches"], 3),
                                 "ms_per_step": round(v["ms"] / reps, 4),
                                 "tflops": round(v["flops"] / (v["ms"] * 1e-3) / 1e12, 2),
                                 "gbytes_per_s": round(v["bytes"] / (v["ms"] * 1e-3) / 1e9, 1)}
                             for k, v in sorted(prof.items(), key=lambda kv: -kv[1]["ms"])}

        y32, yf32 = None, None
        if n_gpus == 1:
            try:
                result["parity_mode"], y32, m32 = parity_mode_leg(model, x, y, args, torch, dev, "bf16x3")
                del m32
            except Exception as e:
                result["parity_mode"] = {"error": f"{type(e).__name__}: {e}"}
            try:
                result["fp32_mode"], yf32, m32 = parity_mode_leg(model, x, y, args, torch, dev, "fp32")
                del m32
            except Exception as e:
                result["fp32_mode"] = {"error": f"{type(e).__name__}: {e}"}
        if not args.no_cpu_baseline and n_gpus == 1:
            # the reference's --device cpu path, restated (oracle), same weights, same batch, fp32
            from oracle import sr_ref as R
            torch.set_num_threads(usable_cores())
            sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
            xc = x.cpu()
            cfg = model.cfg
            with torch.no_grad():
                t0 = time.perf_counter()
                ref = R.drct_forward(sd, xc, cfg)
                first = time.perf_counter() - t0
                n = max(1, min(20, int(args.cpu_seconds / max(first, 1e-3))))
                t0 = time.perf_counter()
                for _ in range(n):
                    R.drct_forward(sd, xc, cfg)
                cpu_t = (time.perf_counter() - t0) / n
            err = float((y.cpu() - ref).abs().max() / ref.abs().max())
            result["cpu_baseline"] = {"value": round(hr_px / cpu_t / 1e6, 4), "unit": "HR Mpixels/s",
                                      "cores": torch.get_num_threads(), "kind": "port",
                                      "sample": f"{n} forwards of the same C2 batch (fp32, torch CPU kernels via "
                                                f"oracle/sr_ref.py), {cpu_t * 1e3:.0f} ms each; {torch.get_num_threads()} threads = "
                                                f"the cores this job may use, host has {os.cpu_count()} logical cores"}
            result["speedup_vs_cpu"] = round(value / (hr_px / cpu_t / 1e6), 1)
            result["max_rel_err_vs_cpu_fp32"] = float(f"{err:.3e}")
            for key, yy in (("parity_mode", y32), ("fp32_mode", yf32)):
                if yy is not None and isinstance(result.get(key), dict) and "error" not in result[key]:
                    e32 = float((yy.cpu() - ref).abs().max() / ref.abs().max())
                    result[key]["max_rel_err_vs_cpu_fp32"] = float(f"{e32:.3e}")
                    result[key]["speedup_vs_cpu"] = round(result[key]["hr_mpixels_per_s"] / (hr_px / cpu_t / 1e6), 1)
        if n_gpus == 1 and not args.no_eval:
            result["anomaly_eval"] = anomaly_eval_leg(model, args, torch)
            try:
                result["scorer"] = scorer_leg(torch, dev)
            except Exception as e:
                result["scorer"] = {"error": f"{type(e).__name__}: {e}"}
            result["eval_1024px_tile"] = c5_leg(args, torch, dev)
            result["drn_forward"] = c3_leg(args, torch, dev)
            if not args.no_train:
                try:
                    result["drn_train"] = drn_train_leg(args, torch, dev)
                except Exception as e:                    # a secondary leg never takes the headline line down
                    result["drn_train"] = {"error": f"{type(e).__name__}: {e}"}
            if not args.no_cpu_baseline:
                try:
                    c3_cpu_baseline(args, torch, dev, result["drn_forward"])
                except Exception as e:
                    result["drn_forward"]["cpu_baseline"] = {"error": f"{type(e).__name__}: {e}"}
        print(json.dumps(result), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def main(argv=None):
    args = parse_args(argv)
    if args.only:
        return run_only(args)
    plan = rank_plan(args)
    if plan:
        # --gpus N without a launcher: start N ranks from THIS process, which has not touched the GPU (no torch.cuda call
        # above); rank 0 prints the JSON line, a failing rank makes the parent exit non-zero
        from srad_amd.launch import spawn
        spawn(run_rank, args.gpus, (args,))
        return
    run_rank(args)


if __name__ == "__main__":
    main()
