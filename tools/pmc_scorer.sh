#!/bin/bash
# HBM-side traffic of the scorer (FETCH_SIZE / WRITE_SIZE, one rocprofv3 pass each - TCC counters cannot share a pass), per
# score_pairs call, for bench.py's two scorer shapes.  Run ON THE GPU BOX from the repo root:  bash tools/pmc_scorer.sh [tag]
# The program sits directly after `--`; counters only.  Output: profiles/<tag>_scorer_pmc.json (+ a copy under gpurun_out/).
set -e -o pipefail
TAG=${1:-r03}
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=gpurun_out/pmc_scorer
mkdir -p "$OUT"
for case in grid_128px tile_1024px; do
  for ctr in FETCH_SIZE WRITE_SIZE; do
    timeout -k 10 300 rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d "$OUT/$case.$ctr" -- python3 tools/scorer_bench.py --case $case --reps 2 > "$OUT/$case.$ctr.log" 2>&1 || { echo "pass $case $ctr failed"; tail -5 "$OUT/$case.$ctr.log"; exit 1; }
    echo "pass $case $ctr done"
  done
done
python3 tools/pmc_scorer.py "profiles/${TAG}_scorer_pmc.json" "$OUT"
cp "profiles/${TAG}_scorer_pmc.json" gpurun_out/
rm -rf "$OUT"
