#!/bin/bash
# After the paired causal blocks and the leaner forward softmax (section 6 row 16): SQ / LDS counters of the training attention kernels (forward + log-sum-exp, flash backward) at 2 x 32 x 2048.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r06/pmc_attn2
mkdir -p $O
export TMPDIR=/tmp
cd /tmp
for c in "GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA" "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_WAVES"; do
  n=$(echo $c | cut -d' ' -f1)
  timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/$n -o a -- python $R/scripts/bench_attn.py > $O/$n.log 2>&1; echo "pmc $n exit $?"
done
cd $R
python scripts/pmc_summary.py $O attn_prefill_kernel attn_bwd_dkv attn_bwd_dq 2>&1 | tee gpurun_out/r06/pmc_attn_after.txt | cut -c1-400
rm -rf $O/*/
