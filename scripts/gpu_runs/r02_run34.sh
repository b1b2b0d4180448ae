R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02_34
mkdir -p $O
export TMPDIR=/tmp
cd $R
LLARK_SK_PLAIN=1 timeout 300 python scripts/bench_streamk.py 2968 2>&1 | grep "bf16 " | grep " o \| down " | tee $O/plain_uniform.log
