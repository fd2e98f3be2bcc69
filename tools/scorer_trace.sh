# per-kernel durations (rocprofv3) of tools/scorer_bench.py
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/sct
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/sct -o w -- python3 $R/tools/scorer_bench.py > /dev/null 2>&1
python3 $R/tools/trace_by_grid.py $R/gpurun_out/sct "_kernel" | grep -E "sat_|ssim_|mse_"
rm -rf $R/gpurun_out/sct
