#!/bin/bash
# Profiling-only builds of the library with one component of the GEMM main loop removed (results invalid).
set -e
cd "$(dirname "$0")/../llark_amd/csrc"
mkdir -p build_ab
for ab in 1 2 3; do
  for f in api vqvae prior llama; do cp build/$f.o build_ab/$f.o 2>/dev/null || true; done
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -DGEMM_ABLATE=$ab -c gemm.hip -o build_ab/gemm_ab$ab.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libllark_hip_ab$ab.so build_ab/api.o build_ab/vqvae.o build_ab/prior.o build_ab/llama.o build_ab/gemm_ab$ab.o
done
ls -la ../libllark_hip_ab*.so
