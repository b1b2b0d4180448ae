# round 2, GPU run 2: MX-MFMA probe (layout / scale / rates), r02 baseline e2e bench + kernel trace, GEMM A/B
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02_2
mkdir -p $O
export TMPDIR=/tmp
cd $R
timeout 120 scripts/probes/mx_probe > $O/mx_probe.txt 2>&1; echo "probe exit $?"; grep -E "^layout fmt=.: (ID|NOT)|^scale|^product|^rate" $O/mx_probe.txt | cut -c1-400
timeout 400 python scripts/bench_gemm256.py 20,30 > $O/bench_gemm256.log 2>&1; echo "bench_gemm256 exit $?"; grep "split f16" $O/bench_gemm256.log
timeout 600 python bench.py --steps 3 --warmup 1 > $O/bench_e2e.log 2>&1; echo "bench exit $?"; tail -1 $O/bench_e2e.log | cut -c1-1800
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof -o r02 -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline > $O/prof_e2e.log 2>&1; echo "rocprof exit $?"
cd $R
python scripts/rocprof_summary.py $O/prof/r02_results.db $O/e2e_kernel_stats.txt; rm -rf $O/prof
head -14 $O/e2e_kernel_stats.txt | cut -c1-170
