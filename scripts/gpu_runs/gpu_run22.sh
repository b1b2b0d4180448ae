mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== full"; timeout 300 python scripts/bench_gemm.py 100 2>&1 | grep -E "fc_qgelu|proj_resid |gate_up|rror" | cut -c1-200
for ab in 11 12 13 14; do
  echo "== ablate $ab (11 no B loads, 12 no A loads+LDS writes, 13 no LDS reads, 14 MFMA only)"; LLARK_SKIP_CHECK=1 LLARK_HIP_LIB=$PWD/llark_amd/libllark_hip_ab$ab.so timeout 300 python scripts/bench_gemm.py 100 2>&1 | grep -E "fc_qgelu|proj_resid |gate_up|rror" | cut -c1-200
done
