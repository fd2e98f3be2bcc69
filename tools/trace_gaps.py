#!/usr/bin/env python3
"""Timeline summary of one training step from a rocprofv3 --kernel-trace CSV: per-queue busy time, idle gaps on the
busiest queue (the data-gradient chain) and overlap between the queues.

    rocprofv3 --kernel-trace --output-format csv -d gpurun_out/trace -- python3 bench.py --steps 5 --warmup 2 --train-steps 4 --no-cpu-baseline --no-eval
    python tools/trace_gaps.py gpurun_out/trace out.json
"""
import collections
import csv
import glob
import json
import sys


def main():
    folder, out = sys.argv[1], sys.argv[2]
    rows = []
    for f in glob.glob(folder + "/**/*kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "0")))
    rows.sort()
    adam = [i for i, r in enumerate(rows) if "adam_kernel" in r[2]]
    # one optimizer step = one adam launch for the SR net; take the span between the last two sync_params launches
    sync = [i for i, r in enumerate(rows) if "sync_params_kernel" in r[2]]
    assert len(sync) >= 3, "need at least three training steps in the trace"
    a, b = sync[-2] + 1, sync[-1] + 1
    step = rows[a:b]
    t0, t1 = step[0][0], max(r[1] for r in step)
    per_q = collections.defaultdict(list)
    for s, e, n, q in step:
        per_q[q].append((s, e, n))
    res = {"step_span_us": (t1 - t0) / 1e3, "kernels": len(step), "queues": {}}
    for q, ks in per_q.items():
        busy = sum(e - s for s, e, _ in ks)
        gaps = [ks[i + 1][0] - ks[i][1] for i in range(len(ks) - 1)]
        pos = [g for g in gaps if g > 0]
        res["queues"][q] = {"launches": len(ks), "busy_us": busy / 1e3, "span_us": (ks[-1][1] - ks[0][0]) / 1e3,
                            "idle_gaps_us": sum(pos) / 1e3, "mean_gap_us": (sum(pos) / max(1, len(pos))) / 1e3,
                            "gaps_over_10us": sum(1 for g in pos if g > 10000),
                            "gap_time_over_10us": sum(g for g in pos if g > 10000) / 1e3}
    by_name = collections.defaultdict(lambda: [0, 0])
    for s, e, n, q in step:
        k = n.replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0].split("<")[0]
        by_name[k][0] += 1
        by_name[k][1] += e - s
    res["by_kernel_us"] = {k: {"launches": v[0], "total_us": v[1] / 1e3, "avg_us": v[1] / v[0] / 1e3}
                           for k, v in sorted(by_name.items(), key=lambda kv: -kv[1][1])}
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps({k: v for k, v in res.items() if k != "by_kernel_us"}, indent=1))
    for k, v in list(res["by_kernel_us"].items())[:24]:
        print(f"  {k:40s} {v['launches']:4d} x {v['avg_us']:7.2f} us = {v['total_us'] / 1e3:6.3f} ms")


if __name__ == "__main__":
    main()
