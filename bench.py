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

PEAK = {"bf16": 2.5e15, "fp32": 157.3e12}      # dense MFMA peaks, MI355X_MICROARCH.md
HBM_PEAK = 8.0e12


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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--batch", type=int, default=4)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world and world > 1:
        print(f"warning: --gpus {args.gpus} but WORLD_SIZE {world}; using WORLD_SIZE", file=sys.stderr)
    n_gpus = world
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the HIP engine has no CPU fallback)")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)

    from srad_amd import _lib as L
    from srad_amd.nets import DRCT

    opt = Opt()
    opt.precision = args.dtype
    opt.use_graph = not args.no_graph
    torch.manual_seed(1)                                   # reference seed (src/main.py:41,89)
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
                   "hipgraph": bool(opt.use_graph)},
    }

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
        traffic, traffic_src = None, None
        try:                                   # PMC bytes per launch of the same workload (tools/pmc_summary.py)
            pmc = json.load(open(os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")))
            traffic = pmc["kernels"][dom]["hbm_bytes_per_launch"]
            traffic_src = "profiles/r01_pmc_traffic.json: " + pmc["source"]
        except Exception:
            pass
        result["roofline"] = {
            "kernel": dom, "bound": "mfma", "achieved": round(achieved / 1e12, 2), "peak": PEAK[args.dtype] / 1e12,
            "unit": "TFLOP/s", "frac": round(achieved / PEAK[args.dtype], 4), "traffic": traffic,
            "traffic_source": traffic_src,
            "launches_per_step": d["launches"] // reps,
            "avg_launch_us": round(d["ms"] * 1e3 / d["launches"], 3),
            "algorithmic_gflop_per_launch": round(d["flops"] / d["launches"] / 1e9, 4),
            "algorithmic_mbytes_per_launch": round(d["bytes"] / d["launches"] / 1e6, 3),
            "share_of_kernel_time": round(d["ms"] / total_ms, 3),
        }
        result["kernels"] = {k: {"launches_per_step": v["launches"] // reps,
                                 "avg_us": round(v["ms"] * 1e3 / v["launches"], 3),
                                 "ms_per_step": round(v["ms"] / reps, 4),
                                 "tflops": round(v["flops"] / (v["ms"] * 1e-3) / 1e12, 2),
                                 "gbytes_per_s": round(v["bytes"] / (v["ms"] * 1e-3) / 1e9, 1)}
                             for k, v in sorted(prof.items(), key=lambda kv: -kv[1]["ms"])}

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
        print(json.dumps(result))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
