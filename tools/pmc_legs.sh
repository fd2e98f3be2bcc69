#!/bin/bash
# Counter passes (FETCH_SIZE | WRITE_SIZE | two SQ groups, tools/pmc_passes.sh) for every bench leg; one summary per leg under
# profiles/<round>_<leg>_pmc_counters.json, which bench.py quotes in that leg's `roofline` while the kernel-source hash matches.
# Run ON THE GPU BOX from the repo root:  bash tools/pmc_legs.sh r03 [legs...]
set -e -o pipefail
R=${1:-r03}; shift || true
LEGS=${@:-"c2 c3 c4 c5 drn_train"}
cd "$(dirname "$0")/.."
export SRAD_BENCH_NO_STEP_GRAPH=1   # eager launches under the counters (bench.py drn_train_leg says why)
for leg in $LEGS; do
  if [ "$leg" = c2 ]; then unset PMC_BENCH; else export PMC_BENCH="python3 bench.py --only $leg"; fi
  bash tools/pmc_passes.sh "gpurun_out/pmc_$leg" "${R}_$leg" > "gpurun_out/pmc_$leg.log" 2>&1 || { echo "leg $leg failed"; tail -5 "gpurun_out/pmc_$leg.log"; exit 1; }
  echo "leg $leg done"
done
