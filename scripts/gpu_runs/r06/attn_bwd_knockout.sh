#!/bin/bash
# What bounds the attention backward: the dQ kernel with parts removed (ATTN_BWD_DEBUG bits: 1 transposing LDS reads, 2 softmax arithmetic, 4 K / V
# fragment reads), per-kernel time from rocprofv3 --kernel-trace at 2 x 32 x 2048.
mkdir -p gpurun_out/r06
out=gpurun_out/r06/attn_bwd_knockout.txt
: > $out
export TMPDIR=/tmp
R=$PWD
for d in 0 1 2 4 5 7; do
  lib=$R/llark_amd/libllark_hip_bwd$d.so
  [ $d == 0 ] && lib=$R/llark_amd/libllark_hip.so
  rm -rf /tmp/kt$d; ( cd /tmp && LLARK_HIP_LIB=$lib timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/kt$d -o a -- python $R/scripts/bench_attn.py > /tmp/kt$d.log 2>&1 )
  f=$(find /tmp/kt$d -name "*.db" | head -1)
  echo "== ATTN_BWD_DEBUG=$d" >> $out
  python scripts/rocprof_summary.py $f /tmp/kt$d.txt > /dev/null 2>&1; grep "attn_bwd_dq\|attn_bwd_dkv" /tmp/kt$d.txt | cut -c1-150 >> $out
done
cat $out
