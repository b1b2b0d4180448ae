#!/bin/bash
# Round 6: why the chained decode is slower -- ring probe + kernel trace of the generate stage with the chain on.
mkdir -p gpurun_out/r06
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
{
timeout 300 python scripts/probes/gemv_ring_probe.py 2>&1 | tail -4
cd /tmp
for ch in 0 1; do
LLARK_DECODE_CHAIN=$ch timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r06/prof_dec$ch -o a -- python $R/bench.py --stages generate --steps 2 --warmup 1 --no-cpu-baseline > $R/gpurun_out/r06/dec_trace$ch.log 2>&1
f=$(find $R/gpurun_out/r06/prof_dec$ch -name "*.db" | head -1); python $R/scripts/rocprof_summary.py $f $R/gpurun_out/r06/decode_chain${ch}_kernel_stats.txt
rm -rf $R/gpurun_out/r06/prof_dec$ch
echo "== chain=$ch"; grep -E "gemv_dma|attn_decode|skinny" $R/gpurun_out/r06/decode_chain${ch}_kernel_stats.txt | cut -c1-60,91-150
done
} 2>&1 | tee $R/gpurun_out/r06/decode_chain_diag.txt
