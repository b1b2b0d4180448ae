#!/bin/bash
# Round 5: SQ / TCC counter passes over the Llama stage (both activation flows in one process) -- where the B-direct kernels' waves wait.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05/pmc_llama_bd
mkdir -p $O
export TMPDIR=/tmp
cd /tmp
i=0
for c in "GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA" \
         "SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU" \
         "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum" ; do
  i=$((i+1))
  timeout 400 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/p$i -o a -- python $R/bench.py --stages llama --steps 2 --warmup 1 --no-cpu-baseline > $O/p$i.log 2>&1; echo "pmc pass $i exit $?"
done
cd $R
python scripts/pmc_summary.py $O gemm_bda_kernel gemm_bd_sk_kernel gemm_bd_kernel attn_prefill rmsnorm > gpurun_out/r05/pmc_llama_bd_summary.txt 2>&1
rm -rf $O/*/
head -c 6000 gpurun_out/r05/pmc_llama_bd_summary.txt
