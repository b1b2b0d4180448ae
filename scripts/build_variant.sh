#!/bin/bash
# Alternate build of the library for same-box A/B runs (LLARK_HIP_LIB=.../libllark_hip_<tag>.so): the listed sources are compiled with the
# extra flags, every other object comes from llark_amd/csrc/build/ (run `make -C llark_amd/csrc` first).
#   bash scripts/build_variant.sh <tag> "<flags>" file1.hip [file2.hip ...]
set -e
tag=$1; flags=$2; shift 2
cd "$(dirname "$0")/../llark_amd/csrc"
mkdir -p build_var/$tag
objs=""
for o in build/*.o; do
  b=$(basename $o .o)
  keep=1
  for f in "$@"; do [ "$b.hip" == "$f" ] && keep=0; done
  [ $keep == 1 ] && objs="$objs $o"
done
for f in "$@"; do
  ( /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wall -Wno-unused-variable -Wno-unused-but-set-variable $flags -c $f -o build_var/$tag/${f%.hip}.o ) &
  objs="$objs build_var/$tag/${f%.hip}.o"
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libllark_hip_$tag.so $objs
ls -la ../libllark_hip_$tag.so
