R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02_18
mkdir -p $O
export TMPDIR=/tmp
cd $R
for i in 1 2; do
timeout 400 python scripts/bench_gemm256.py 41 > $O/bench_n_prio_$i.log 2>&1; echo "prio run $i"; grep "split f16" $O/bench_n_prio_$i.log | cut -c1-110
LLARK_HIP_LIB=$R/llark_amd/libllark_hip_n_noprio.so timeout 400 python scripts/bench_gemm256.py 41 > $O/bench_n_noprio_$i.log 2>&1; echo "noprio run $i"; grep "split f16" $O/bench_n_noprio_$i.log | cut -c1-110
done
timeout 900 python bench.py --no-cpu-baseline > $O/bench_e2e.log 2>&1; echo "e2e exit $?"; tail -1 $O/bench_e2e.log | cut -c1-330
