mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python bench.py --stages mpt-train --steps 3 --warmup 1 > gpurun_out/bench_mpt_train.log 2>&1; echo "exit $?"; tail -c 900 gpurun_out/bench_mpt_train.log
