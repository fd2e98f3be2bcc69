# per-kernel durations (rocprofv3) of one DRN-L training step (tools/drn_train_bench.py), per grid size: weight-gradient kernels
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/drtt
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/drtt -o w -- python3 $R/tools/drn_train_bench.py --steps 3 > /dev/null 2>&1
python3 $R/tools/trace_by_grid.py $R/gpurun_out/drtt "${1:-wgrad}"
rm -rf $R/gpurun_out/drtt
