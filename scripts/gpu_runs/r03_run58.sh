#!/bin/bash
# PMC passes (counters only + kernel trace) of the round's new training kernels on the shipped library: attention forward / backward
# (scripts/bench_attn.py) and llark_gemm16_t on one layer's dX / dW products (scripts/bench_gemm_train.py 200 4096)
R=$GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$R/gpurun_out/r03_pmc_train_kernels
mkdir -p $O
cd /tmp
C1="GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA"
C2="SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_MFMA"
i=0
for c in "$C1" "$C2"; do
  i=$((i+1))
  timeout 200 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/attn_p$i -o a -- python $R/scripts/bench_attn.py > $O/attn_p$i.log 2>&1; echo "attn pass $i exit $?"
  timeout 200 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/gemm_p$i -o g -- python $R/scripts/bench_gemm_train.py 200 4096 > $O/gemm_p$i.log 2>&1; echo "gemm pass $i exit $?"
done
cd $R
mkdir -p $O/attn $O/gemm; mv $O/attn_p1 $O/attn_p2 $O/attn/; mv $O/gemm_p1 $O/gemm_p2 $O/gemm/
python scripts/pmc_summary.py $O/attn attn_ > gpurun_out/r03_pmc_attn_final.txt 2>&1; cat gpurun_out/r03_pmc_attn_final.txt | cut -c1-260
python scripts/pmc_summary.py $O/gemm gemm_t_kernel > gpurun_out/r03_pmc_gemm_tn.txt 2>&1; cat gpurun_out/r03_pmc_gemm_tn.txt | cut -c1-260
tail -3 $O/gemm_p2.log
find $O -name "*.csv" -size +20M -delete; du -sh $O
