#!/bin/bash
# Profiling-only builds of the library with one component of the GEMM main loop removed (results invalid).
# ABLATIONS="11 12 13 14" bash scripts/build_ablations.sh   (parallel builds)
set -e
cd "$(dirname "$0")/../llark_amd/csrc"
mkdir -p build_ab
for ab in ${ABLATIONS:-1 2 3}; do
  ( /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -DGEMM_ABLATE=$ab -c gemm.hip -o build_ab/gemm_ab$ab.o &&
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libllark_hip_ab$ab.so build/api.o build/vqvae.o build/prior.o build/llama.o build/train.o build_ab/gemm_ab$ab.o ) &
done
wait
ls -la ../libllark_hip_ab*.so
