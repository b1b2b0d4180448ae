mkdir -p gpurun_out
export TMPDIR=/tmp
for a in decode flash; do echo "LLARK_DECODE_ATTN=$a"; LLARK_DECODE_ATTN=$a timeout 600 python scripts/bench_kernels.py decode 2>&1 | grep "eager"; done
