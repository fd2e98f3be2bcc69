set -o pipefail
R=$GRAFT_REPO_ROOT
cd $R
bash tools/pmc_passes.sh gpurun_out/pmc_final r02 > gpurun_out/pmc_final.log 2>&1 || { tail -5 gpurun_out/pmc_final.log; exit 1; }
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/kstats
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/kstats -o ks -- python3 $R/bench.py --no-cpu-baseline > $R/gpurun_out/kstats.log 2>&1 || { tail -5 $R/gpurun_out/kstats.log; exit 1; }
cd $R
find gpurun_out/kstats -name "*kernel_stats.csv" -exec cp {} gpurun_out/r02_c_kernel_stats.csv \;
rm -rf gpurun_out/kstats
timeout -k 10 420 python bench.py > gpurun_out/bench_final.log 2>&1 || { tail -5 gpurun_out/bench_final.log; exit 1; }
tail -c 300 gpurun_out/bench_final.log
ls -la gpurun_out/pmc_final.counters.json gpurun_out/r02_c_kernel_stats.csv
