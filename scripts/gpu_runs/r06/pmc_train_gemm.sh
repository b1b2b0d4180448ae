#!/bin/bash
# Round 6 (VERDICT r05 item 1c "done = --pmc shows pipe busy >= 65 %"): SQ / GRBM counter pass over the training products at M = 4096:
# llark_gemm16_t (one LDS stage, rounds 3-5: variant 210) against the dW form on the DMA loop (llark_gemm16_ta_fragw: variant 220) and the
# fragment-major B-direct products the forward / dX take (variant 102).
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r06/pmc_train
mkdir -p $O
export TMPDIR=/tmp
cd /tmp
for c in "GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA" "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAIT_INST_LDS"; do
  n=$(echo $c | cut -d' ' -f1)
  timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/$n -o a -- python $R/scripts/bench_gemm_train.py 210,220,102 4096 > $O/$n.log 2>&1; echo "pmc $n exit $?"
done
cd $R
python scripts/pmc_summary.py $O gemm_t_kernel gemm_bda_ta_kernel gemm_bda_kernel 2>&1 | tee gpurun_out/r06/pmc_train_gemm.txt | cut -c1-260
rm -rf $O/*/
