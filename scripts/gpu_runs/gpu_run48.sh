mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python bench.py --stages mpt --steps 2 --warmup 1 > gpurun_out/bench_mpt.log 2>&1; echo "mpt exit $?"; tail -c 1300 gpurun_out/bench_mpt.log
timeout 600 python bench.py --stages mpt --llm-precision bf16 --steps 2 --warmup 1 > gpurun_out/bench_mpt_bf16.log 2>&1; echo "mpt bf16 exit $?"; grep -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*' gpurun_out/bench_mpt_bf16.log
timeout 900 python bench.py --steps 2 --warmup 1 > gpurun_out/bench_e2e.log 2>&1; echo "e2e exit $?: $(grep -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*' gpurun_out/bench_e2e.log | tr '\n' ' ')"
