#!/bin/bash
# A/B builds of the decode (skinny) GEMM: non-temporal weight loads and 8 k-steps in flight.  Only gemm.hip is recompiled.
set -e
cd "$(dirname "$0")/../llark_amd/csrc"
mkdir -p build_ab
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off"
OBJS=$(ls build/*.o | grep -v "build/gemm.o")
for v in "nt:-DSKINNY_NT=1" "d8:-DSKINNY_DEPTH=8" "ntd8:-DSKINNY_NT=1 -DSKINNY_DEPTH=8"; do
  name=${v%%:*}; defs=${v#*:}
  ( /opt/rocm/bin/hipcc $FLAGS $defs -c gemm.hip -o build_ab/gemm_sk_$name.o && /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libllark_hip_sk_$name.so $OBJS build_ab/gemm_sk_$name.o ) &
done
wait
ls -la ../libllark_hip_sk_*.so
