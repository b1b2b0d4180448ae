#!/bin/bash
# Profiling-only build of the library with per-phase cycle counters in the 256x256 GEMM tiles (results valid, timing perturbed).
set -e
cd "$(dirname "$0")/../llark_amd/csrc"
mkdir -p build_ab
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off"
/opt/rocm/bin/hipcc $FLAGS -DLLARK_LO8_PROF ${EXTRA_DEFS} -c gemm256.hip -o build_ab/gemm256_prof.o &
/opt/rocm/bin/hipcc $FLAGS -DLLARK_LO8_PROF ${EXTRA_DEFS} -c gemm256n.hip -o build_ab/gemm256n_prof.o &
/opt/rocm/bin/hipcc $FLAGS -DLLARK_LO8_PROF ${EXTRA_DEFS} -c gemm256_lo8n.hip -o build_ab/gemm256_lo8n_prof.o &
wait
OBJS=$(ls build/*.o | grep -v "gemm256.o\|gemm256n.o\|gemm256_lo8n.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libllark_hip_lo8prof.so $OBJS build_ab/gemm256_prof.o build_ab/gemm256n_prof.o build_ab/gemm256_lo8n_prof.o
ls -la ../libllark_hip_lo8prof.so
