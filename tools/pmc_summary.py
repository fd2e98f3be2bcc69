#!/usr/bin/env python3
"""Summarise rocprofv3 PMC passes of the C2 forward into profiles/rNN_pmc_counters.json, which bench.py quotes as
roofline.traffic / roofline.mfma_busy while its `kernel_source_sha` matches the kernels being benched.

Passes (tools/pmc_passes.sh runs them; each its own process, counters only with --kernel-trace, as the pool requires):
  traffic : FETCH_SIZE | WRITE_SIZE            (TCC: cannot share a pass, MI355X_MICROARCH.md "rocprofv3 PMC slots")
  sq      : SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES
            SQ_INSTS_VALU_MFMA_MOPS_BF16 (or the names `rocprofv3 -L` lists), SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
            SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU, GRBM_GUI_ACTIVE
Every <dir> given on the command line is scanned for *counter_collection.csv (Counter_Name / Counter_Value per dispatch)
and *kernel_trace.csv (durations); counters are averaged per launch and per kernel class (and per template instance).

    python tools/pmc_summary.py profiles/r02_pmc_counters.json gpurun_out/pmc_*/

gfx950 corrections (MI355X_MICROARCH.md): FETCH_SIZE / WRITE_SIZE are KiB; FETCH_SIZE tallies the 128-B requests of wide
(16 B / lane) coalesced reads at 64 B, so it is doubled - every global load of qkv_attn / mlp_block is a 16-byte-per-lane
load of whole 64-byte row pieces or 1 KB weight fragments, the pattern the guide's factor was measured on; narrower accesses
are uncalibrated.  SQ_*_CYCLES count quad-cycles (x4 = shader cycles) except SQ_VALU_MFMA_BUSY_CYCLES (cycles, summed over
the SIMDs that were busy); SQ_BUSY_CYCLES / GRBM_GUI_ACTIVE are summed over the 8 XCDs (per shader engine for SQ).

Derived per kernel class:
  mfma_busy            = SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMDs x 256 CUs x kernel cycles), kernel cycles = SQ_BUSY_CYCLES / 32
                         (share of all matrix pipes' time spent in MFMA while the kernel runs: the north_star's "MFMA-busy")
  wait_frac            = SQ_WAIT_ANY / SQ_WAVE_CYCLES          (waves parked at s_waitcnt / s_barrier)
  issue_stall_frac     = SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES
  active_frac          = SQ_ACTIVE_INST_ANY / SQ_WAVE_CYCLES
  lds_bank_conflict_frac = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE
"""
import collections
import csv
import glob
import hashlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "anomaly-detection-super-resolution_amd", "csrc")


def kernel_source_sha() -> str:
    h = hashlib.sha256()
    for name in sorted(os.listdir(CSRC)):
        if name.endswith((".hip", ".h")):
            h.update(name.encode())
            h.update(open(os.path.join(CSRC, name), "rb").read())
    return h.hexdigest()[:16]


def classify(name: str) -> str:
    if "gemm_kernel" in name:
        args = name.split("gemm_kernel<")[1].split(">")[0].replace(" ", "").split(",")
        return {"64": "gemm_bn64", "32": "gemm_bn32", "16": "gemm_bn16"}.get(args[2], "gemm")
    for key, cls in (("mlp_block_kernel", "mlp_block"), ("qkv_attn_kernel", "qkv_attn"), ("swin_block_kernel", "swin_block"),
                     ("window_attn_bwd", "window_attn_bwd"), ("window_attn_kernel", "window_attn"), ("layernorm_kernel", "layernorm"),
                     ("wgrad_multi", "wgrad"), ("conv80_kernel", "conv80"), ("wgrad_kernel", "wgrad"), ("wgrad_reduce_kernel", "wgrad_reduce"),
                     ("ln_bwd_kernel", "layernorm_bwd"), ("mlp_bwd_kernel", "mlp_bwd"), ("lin_ln_bwd_kernel", "lin_ln_bwd"),
                     ("ln_qkv_kernel", "ln_qkv"), ("wgrad_conv9", "wgrad"), ("wgrad80", "wgrad"), ("adam", "optim"), ("pack_", "pack_weight"), ("sync_params", "pack_weight"),
                     ("sat_", "scorer"), ("ssim_", "scorer"), ("mse_", "scorer")):
        if key in name:
            return cls
    return "other"


def instance(name: str) -> str:
    n = name.replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "")
    return n.split("(")[0][:80]


def main():
    dst, dirs = sys.argv[1], sys.argv[2:]
    cls_sum = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))     # class -> counter -> [n, sum]
    inst_sum = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
    dur = collections.defaultdict(lambda: [0, 0.0])
    for d in dirs:
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                for table, key in ((cls_sum, classify(r["Kernel_Name"])), (inst_sum, instance(r["Kernel_Name"]))):
                    e = table[key][r["Counter_Name"]]
                    e[0] += 1
                    e[1] += float(r["Counter_Value"])
        for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                e = dur[classify(r["Kernel_Name"])]
                e[0] += 1
                e[1] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])

    def summarise(ctrs):
        avg = {k: v[1] / v[0] for k, v in ctrs.items() if v[0]}
        out = {"launches": max(v[0] for v in ctrs.values())}
        out["counters_per_launch"] = {k: round(v, 1) for k, v in sorted(avg.items())}
        if "FETCH_SIZE" in avg:
            out["fetch_bytes_per_launch"] = round(avg["FETCH_SIZE"] * 1024 * 2)
        if "WRITE_SIZE" in avg:
            out["write_bytes_per_launch"] = round(avg["WRITE_SIZE"] * 1024)
        if "FETCH_SIZE" in avg and "WRITE_SIZE" in avg:
            out["hbm_bytes_per_launch"] = out["fetch_bytes_per_launch"] + out["write_bytes_per_launch"]
        wave = avg.get("SQ_WAVE_CYCLES")
        if wave:
            for name, key in (("wait_frac", "SQ_WAIT_ANY"), ("issue_stall_frac", "SQ_WAIT_INST_ANY"), ("active_frac", "SQ_ACTIVE_INST_ANY"),
                              ("valu_active_frac", "SQ_ACTIVE_INST_VALU"), ("lds_active_frac", "SQ_ACTIVE_INST_LDS")):
                if key in avg:
                    out[name] = round(avg[key] / wave, 4)
        if avg.get("SQ_LDS_IDX_ACTIVE"):
            out["lds_bank_conflict_frac"] = round(avg.get("SQ_LDS_BANK_CONFLICT", 0.0) / avg["SQ_LDS_IDX_ACTIVE"], 4)
        # kernel length in shader cycles: SQ_BUSY_CYCLES is summed over the 32 shader engines and counts only while waves are
        # resident (15.7 us dispatches read ~32 k cycles = 2.1 GHz); GRBM_GUI_ACTIVE / 8 also counts the dispatch ramp of the
        # profiled launch and reads ~60 % high on kernels this short (MI355X_MICROARCH.md "DVFS give-back"), so it is the fallback
        kcycles = None
        if avg.get("SQ_BUSY_CYCLES", 0) > 0:
            kcycles = avg["SQ_BUSY_CYCLES"] / 32.0
        elif avg.get("GRBM_GUI_ACTIVE", 0) > 0:
            kcycles = avg["GRBM_GUI_ACTIVE"] / 8.0
        if kcycles:
            out["kernel_cycles"] = round(kcycles)
            if "SQ_VALU_MFMA_BUSY_CYCLES" in avg:
                out["mfma_busy"] = round(avg["SQ_VALU_MFMA_BUSY_CYCLES"] / (4 * 256 * kcycles), 4)
                out["mfma_busy_note"] = "SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x SQ_BUSY_CYCLES / 32): share of the chip's matrix-pipe time while the kernel is resident"
            if "GRBM_GUI_ACTIVE" in avg and avg["GRBM_GUI_ACTIVE"] > 0:
                out["grbm_cycles"] = round(avg["GRBM_GUI_ACTIVE"] / 8.0)
        return out

    res = {"source": "rocprofv3 --pmc passes (one counter group per process) of " + os.environ.get("PMC_BENCH", "`bench.py --no-graph --no-train "
                     "--no-eval --no-cpu-baseline`: the C2 forward") + ", eager launches",
           "corrections": "FETCH_SIZE/WRITE_SIZE KiB -> bytes; FETCH_SIZE x2 (16 B/lane coalesced loads: 128-B requests tallied at 64 B)",
           "kernel_source_sha": kernel_source_sha(), "kernels": {}, "instances": {}}
    for k, ctrs in cls_sum.items():
        res["kernels"][k] = summarise(ctrs)
        if dur[k][0]:
            res["kernels"][k]["dispatch_us"] = round(dur[k][1] / dur[k][0] / 1e3, 3)
    for k, ctrs in inst_sum.items():
        if any(s in k for s in ("qkv_attn", "mlp_block", "swin_block", "window_attn")):
            res["instances"][k] = summarise(ctrs)
    os.makedirs(os.path.dirname(os.path.abspath(dst)), exist_ok=True)
    json.dump(res, open(dst, "w"), indent=1)
    keep = {k: {kk: vv for kk, vv in v.items() if kk != "counters_per_launch"} for k, v in res["kernels"].items()}
    print(json.dumps(keep, indent=1))


if __name__ == "__main__":
    main()
