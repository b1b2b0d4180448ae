R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02_55
mkdir -p $O
export TMPDIR=/tmp
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof -o r02 -- python $R/bench.py --stages generate --steps 1 --warmup 1 --no-cpu-baseline > $O/prof_generate.log 2>&1; echo "rocprof exit $?"
cd $R
python scripts/rocprof_summary.py $O/prof/r02_results.db $O/generate_kernel_stats.txt; rm -rf $O/prof
head -14 $O/generate_kernel_stats.txt | cut -c1-170
