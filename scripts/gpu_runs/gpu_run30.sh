mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_prior_gpu.py -m gpu -q --tb=short -x -p no:cacheprovider -k "persistent" > gpurun_out/tests30.log 2>&1; echo "tests exit $?"
grep -E "passed|failed" gpurun_out/tests30.log | tail -2; grep -E "^E  " gpurun_out/tests30.log | cut -c1-300 | head -20
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 1 --warmup 1 --depth 2 --no-cpu-baseline > gpurun_out/bench_torchrun.log 2>&1; echo "torchrun bench exit $?"; tail -c 600 gpurun_out/bench_torchrun.log
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29512 bench.py --gpus 1 --stages train --llm-layers 2 --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/bench_torchrun_train.log 2>&1; echo "torchrun train exit $?"; tail -c 400 gpurun_out/bench_torchrun_train.log
