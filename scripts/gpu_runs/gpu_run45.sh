export TMPDIR=/tmp
echo "== full"; timeout 300 python scripts/bench_kernels.py res 2>&1 | grep resblock
for ab in 1 2; do echo "== ablate $ab (1 no global loads, 2 no MFMA)"; LLARK_HIP_LIB=$PWD/llark_amd/libllark_hip_res$ab.so timeout 300 python scripts/bench_kernels.py res 2>&1 | grep resblock; done
