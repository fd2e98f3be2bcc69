#!/usr/bin/env python3
"""Summarise rocprofv3 PMC passes (one per counter, as MI355X_MICROARCH.md prescribes) into a small
JSON that bench.py reports as roofline.traffic.

    rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/prof_fetch -- python3 bench.py --steps 5 --warmup 3 --no-cpu-baseline --no-graph
    rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/prof_write -- python3 bench.py ...
    python tools/pmc_summary.py gpurun_out/prof_fetch gpurun_out/prof_write profiles/r01_pmc_traffic.json

gfx950 corrections: FETCH_SIZE / WRITE_SIZE are in KiB; FETCH_SIZE counts 128-B requests as 64 B for
wide coalesced reads, so it is doubled (MI355X_MICROARCH.md "HBM").
"""
import collections
import csv
import glob
import json
import sys


def classify(name: str) -> str:
    if "gemm_kernel" in name:
        args = name.split("gemm_kernel<")[1].split(">")[0].replace(" ", "").split(",")
        return {"64": "gemm_bn64", "32": "gemm_bn32", "16": "gemm_bn16"}[args[2]]
    if "mlp_block_kernel" in name:
        return "mlp_block"
    if "qkv_attn_kernel" in name:
        return "qkv_attn"
    if "window_attn_kernel" in name:
        return "window_attn"
    if "layernorm_kernel" in name:
        return "layernorm"
    for key, cls in (("wgrad_multi_kernel", "wgrad"), ("wgrad_kernel", "wgrad"), ("wgrad_reduce_kernel", "wgrad_reduce"),
                     ("ln_bwd_kernel", "layernorm_bwd"), ("window_attn_bwd", "window_attn_bwd")):
        if key in name:
            return cls
    return "other"


def load(folder):
    out = collections.defaultdict(lambda: [0, 0.0])
    for f in glob.glob(folder + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = classify(r["Kernel_Name"])
            out[k][0] += 1
            out[k][1] += float(r["Counter_Value"])
    return out


def main():
    fetch, write, dst = load(sys.argv[1]), load(sys.argv[2]), sys.argv[3]
    res = {"source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), bench.py --no-graph C2 workload",
           "corrections": "KiB -> bytes; FETCH_SIZE x2 on gfx950 (128-B requests tallied at 64 B)", "kernels": {}}
    for k in fetch:
        n = fetch[k][0]
        f = fetch[k][1] / n * 1024 * 2
        w = write[k][1] / max(1, write[k][0]) * 1024 if k in write else 0.0
        res["kernels"][k] = {"launches": n, "fetch_bytes_per_launch": round(f), "write_bytes_per_launch": round(w),
                             "hbm_bytes_per_launch": round(f + w)}
    json.dump(res, open(dst, "w"), indent=1)
    print(json.dumps(res["kernels"], indent=1))


if __name__ == "__main__":
    main()
