mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 300 python bench.py --stages train --llm-layers 2 --steps 2 --warmup 1 > gpurun_out/bench_train_l2.log 2>&1; echo "train l2 exit $?"; tail -c 1500 gpurun_out/bench_train_l2.log
timeout 600 python bench.py --stages train --steps 3 --warmup 1 > gpurun_out/bench_train.log 2>&1; echo "train exit $?"; tail -c 2500 gpurun_out/bench_train.log
