#!/bin/bash
# Counter passes of the C2 forward for tools/pmc_summary.py.  Run ON THE GPU BOX from the repo root:
#     bash tools/pmc_passes.sh [out_dir] [tag]
# One rocprofv3 process per counter group (TCC counters cannot share a pass; SQ has 8 slots), the program directly after
# `--` (no env / bash -c hop: the profiler's preload has already initialised the GPU), counters only (no sys/hip traces).
set -e -o pipefail
OUT=${1:-gpurun_out/pmc}
TAG=${2:-r02}
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
BENCH=${PMC_BENCH:-"python3 bench.py --steps 3 --warmup 3 --no-graph --no-train --no-eval --no-cpu-baseline"}   # PMC_BENCH="python3 tools/x.py": another program
run() {  # name, counters...
  local name=$1; shift
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d "$OUT/$name" -- $BENCH > "$OUT.$name.log" 2>&1 || { echo "pass $name failed"; tail -5 "$OUT.$name.log"; return 1; }
  echo "pass $name done"
}
mkdir -p "$OUT"
run fetch FETCH_SIZE
run write WRITE_SIZE
run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE
run sq2 SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_WAVES GRBM_GUI_ACTIVE
python3 tools/pmc_summary.py "profiles/${TAG}_pmc_counters.json" "$OUT"/fetch "$OUT"/write "$OUT"/sq1 "$OUT"/sq2 > "$OUT.summary.txt"
cp "profiles/${TAG}_pmc_counters.json" "$OUT.counters.json"
rm -rf "$OUT"   # the raw per-dispatch CSVs are tens of MB: only the summaries travel back
echo "summary written"
