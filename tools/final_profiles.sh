#!/bin/bash
# The tracked evidence of a round at the CURRENT kernel sources.  Run ON THE GPU BOX from the repo root (two gpurun calls fit
# the 20-minute limit):   bash tools/final_profiles.sh r03 pmc     then     bash tools/final_profiles.sh r03 bench
#   pmc:   counter passes per bench leg (tools/pmc_legs.sh) and for the scorer (tools/pmc_scorer.sh)
#          -> gpurun_out/<round>_<leg>_pmc_counters.json, <round>_scorer_pmc.json      (copy them into profiles/ and commit)
#   bench: rocprofv3 --kernel-trace --stats of the default bench command -> gpurun_out/<round>_kernel_stats.csv, then the
#          default bench line (which quotes the counter summaries found under profiles/) -> gpurun_out/<round>_bench.json
set -e -o pipefail
R=${1:-r03}; WHAT=${2:-pmc}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOT"
if [ "$WHAT" = pmc ]; then
  shift; shift || true
  LEGS=${@:-"c2 c3 c4 c5 drn_train"}
  bash tools/pmc_legs.sh "$R" $LEGS
  for leg in $LEGS; do cp "profiles/${R}_${leg}_pmc_counters.json" gpurun_out/; done
  bash tools/pmc_scorer.sh "$R"
  ls -la gpurun_out/${R}_*pmc*.json
else
  cd /tmp && export TMPDIR=/tmp
  rm -rf "$ROOT/gpurun_out/kstats"
  timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d "$ROOT/gpurun_out/kstats" -o ks -- python3 "$ROOT/bench.py" --no-cpu-baseline > "$ROOT/gpurun_out/kstats.log" 2>&1 || { tail -5 "$ROOT/gpurun_out/kstats.log"; exit 1; }
  cd "$ROOT"
  find gpurun_out/kstats -name "*kernel_stats.csv" -exec cp {} "gpurun_out/${R}_kernel_stats.csv" \;
  rm -rf gpurun_out/kstats
  timeout -k 10 500 python3 bench.py > gpurun_out/bench_final.log 2>&1 || { tail -5 gpurun_out/bench_final.log; exit 1; }
  tail -n 1 gpurun_out/bench_final.log > "gpurun_out/${R}_bench.json"
  tail -c 400 gpurun_out/bench_final.log
  ls -la "gpurun_out/${R}_kernel_stats.csv" "gpurun_out/${R}_bench.json"
fi
