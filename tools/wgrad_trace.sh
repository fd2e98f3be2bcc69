# kernel durations (rocprofv3) of the weight-gradient kernels in tools/wgrad_bench.py, per grid size (= per block width)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/wgt
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/wgt -o w -- python3 $R/tools/wgrad_bench.py > /dev/null 2>&1
python3 $R/tools/trace_by_grid.py $R/gpurun_out/wgt wgrad_
rm -rf $R/gpurun_out/wgt
