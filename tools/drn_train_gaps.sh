# timeline of one DRN-L training step (tools/drn_train_bench.py): per-queue busy time, idle gaps, kernels by total time
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/drtg
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/drtg -o w -- python3 $R/tools/drn_train_bench.py --steps 3 > /dev/null 2>&1
python3 $R/tools/trace_gaps.py $R/gpurun_out/drtg $R/gpurun_out/drn_train_gaps.json
rm -rf $R/gpurun_out/drtg
