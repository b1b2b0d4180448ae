mkdir -p gpurun_out
export TMPDIR=/tmp
for rep in 1 2; do for pz in 0 1; do
LLARK_GEMM_PERSIST=$pz timeout 600 python bench.py --stages jukebox --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/bench_jb_p$pz.log 2>&1; echo "persist=$pz: $(grep -o '"ms_per_step": [0-9.]*' gpurun_out/bench_jb_p$pz.log) $(grep -o '"gemm_split_f16": [0-9.]*' gpurun_out/bench_jb_p$pz.log)"
done; done
