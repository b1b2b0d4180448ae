#!/bin/bash
# round 4, run 9: prior attention on the 16-bit matrix cores (prior_attn16_kernel): parity, per-pattern timing, bank conflicts, e2e
mkdir -p gpurun_out/r04
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
( timeout 900 python -m pytest tests/test_prior_gpu.py tests/test_lo8_gpu.py tests/test_fulldepth_gpu.py -x -q -s -k "attention or prior or jukebox or lo8" 2>&1 | grep -E "fulldepth|passed|failed|Error|error|attn pattern" | tail -30 ) > gpurun_out/r04/run9_tests.txt
( timeout 300 python scripts/bench_kernels.py attn 2>&1 | tail -4 ) > gpurun_out/r04/attn16_bench.txt
( timeout 300 rocprofv3 --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAVE_CYCLES --kernel-trace --output-format csv -d gpurun_out/r04/pmc_attn16 -o a -- python scripts/bench_kernels.py attn > /dev/null 2>&1
  python - <<'PY'
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/r04/pmc_attn16/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "prior_attn" in row["Kernel_Name"]:
            agg[row["Kernel_Name"][:50] + "|grid" + row["Grid_Size"]][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, cs in agg.items():
    print(k, {c: round(sum(v) / len(v)) for c, v in cs.items()})
PY
) > gpurun_out/r04/attn16_pmc.txt 2>&1
rm -rf gpurun_out/r04/pmc_attn16
( timeout 900 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras --no-alt-precision 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('e2e ms/step', d['ms_per_step'], 'clips/s', d['value'], 'gemm frac', d['roofline']['frac'], d['kernel_ms'])" ) > gpurun_out/r04/run9_bench.txt
cat gpurun_out/r04/run9_tests.txt gpurun_out/r04/attn16_bench.txt gpurun_out/r04/attn16_pmc.txt gpurun_out/r04/run9_bench.txt
