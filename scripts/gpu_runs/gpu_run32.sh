mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
LLARK_DECODE_GRAPH=0 timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_dec -o r -- python $R/scripts/bench_kernels.py decode > $R/gpurun_out/prof_dec.log 2>&1; echo "exit $?"
python $R/scripts/rocprof_summary.py $(find $R/gpurun_out/prof_dec -name "*.db" | head -1) > $R/gpurun_out/prof_dec_stats.txt
head -30 $R/gpurun_out/prof_dec_stats.txt | cut -c1-180
rm -rf $R/gpurun_out/prof_dec
