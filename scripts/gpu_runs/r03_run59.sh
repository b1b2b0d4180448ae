#!/bin/bash
# conflict-free LDS layout of the attention tiles: parity tests, micro-benchmark, one PMC pass (LDS counters)
R=$GRAFT_REPO_ROOT
cd $R && export TMPDIR=/tmp
( timeout 400 python -m pytest tests/test_attn_bwd_gpu.py tests/test_llama_gpu.py tests/test_mpt_gpu.py -q -m gpu -x 2>&1 | tail -3 ) > gpurun_out/r03_attn_swz_tests.txt; cat gpurun_out/r03_attn_swz_tests.txt
timeout 200 python scripts/bench_attn.py 2>&1 | tail -8 > gpurun_out/r03_bench_attn_v3.txt; cat gpurun_out/r03_bench_attn_v3.txt
O=$R/gpurun_out/r03_pmc_attn_swz
mkdir -p $O; cd /tmp
C1="GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA"
C2="SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_MFMA"
timeout 200 rocprofv3 --pmc $C1 --kernel-trace --output-format csv -d $O/p1 -o a -- python $R/scripts/bench_attn.py > $O/p1.log 2>&1; echo "pass 1 exit $?"
timeout 200 rocprofv3 --pmc $C2 --kernel-trace --output-format csv -d $O/p2 -o a -- python $R/scripts/bench_attn.py > $O/p2.log 2>&1; echo "pass 2 exit $?"
cd $R; python scripts/pmc_summary.py $O attn_ > gpurun_out/r03_pmc_attn_swz.txt 2>&1; grep -A2 "grid 524288\|2, false.*grid 262144" gpurun_out/r03_pmc_attn_swz.txt | grep -v "^   [A-Z]" | cut -c1-250
rm -rf $O
