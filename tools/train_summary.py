#!/usr/bin/env python3
"""Print the training leg of a bench.py JSON line: python tools/train_summary.py gpurun_out/bench_train.log"""
import json
import sys

r = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
t = r["train"]
print("ms/step", t["ms_per_step"], "images/s", t["images_per_s"], "TFLOP/s", t.get("model_tflops"), "loss", t["loss_first_last"])
for k, v in t.get("kernels", {}).items():
    print(f"  {k:18s} {v}")
