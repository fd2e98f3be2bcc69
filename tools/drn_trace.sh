# per-kernel durations (rocprofv3) of tools/drn_bench.py, per grid size
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/drt
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/drt -o w -- python3 $R/tools/drn_bench.py --steps 5 > /dev/null 2>&1
python3 $R/tools/trace_by_grid.py $R/gpurun_out/drt "" | grep -E "conv80|gemm_kernel<1, (64|128), 80" 
rm -rf $R/gpurun_out/drt
