#!/bin/bash
# Profiling builds of the prior attention kernel with one stage removed each (results invalid, timing only).
set -e
cd "$(dirname "$0")/../llark_amd/csrc"
mkdir -p build_ab
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off"
OBJS=$(ls build/*.o | grep -v "build/prior.o")
for v in 1 2 3 4; do
  ( /opt/rocm/bin/hipcc $FLAGS -DATTN_ABLATE=$v -c prior.hip -o build_ab/prior_ab$v.o && /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libllark_hip_attn_ab$v.so $OBJS build_ab/prior_ab$v.o ) &
done
wait
ls -la ../libllark_hip_attn_ab*.so
