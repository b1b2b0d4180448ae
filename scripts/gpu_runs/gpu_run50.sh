export TMPDIR=/tmp
mkdir -p gpurun_out
for f in 1 0 1 0; do LLARK_DECODE_FUSE_NORM_A=$f timeout 600 python bench.py --stages generate --no-cpu-baseline --steps 3 > gpurun_out/bench_gen_f$f.log 2>&1; echo "fuse_norm_a=$f: $(grep -o '"ms_per_step": [0-9.]*\|"decode_ms_per_token": [0-9.]*' gpurun_out/bench_gen_f$f.log | tr '\n' ' ')"; done
