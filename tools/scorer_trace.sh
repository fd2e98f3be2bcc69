#!/bin/bash
# Per-kernel times of the scorer at bench.py's two shapes (rocprofv3 kernel trace of tools/scorer_bench.py).  Run on the GPU box.
set -e -o pipefail
OUT=${1:-gpurun_out/scorer_trace}
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
mkdir -p "$OUT"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT" -- python3 tools/scorer_bench.py > "$OUT.log" 2>&1
f=$(find "$OUT" -name "*kernel_stats.csv" | head -1)
cp "$f" "$OUT.kernel_stats.csv"
rm -rf "$OUT"
head -12 "$OUT.kernel_stats.csv"
