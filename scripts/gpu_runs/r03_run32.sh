#!/bin/bash
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03_pmc_pattn
mkdir -p $O
export TMPDIR=/tmp
cd /tmp
for c in "FETCH_SIZE" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum" "GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA" "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"; do
  n=$(echo $c | cut -d' ' -f1)
  timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/$n -o a -- python $R/scripts/bench_kernels.py attn > $O/$n.log 2>&1; echo "pmc $n exit $?"
done
cd $R
python - <<'PY' | tee gpurun_out/r03_pmc_prior_attn_summary.txt
import csv, glob, collections
for f in sorted(glob.glob("gpurun_out/r03_pmc_pattn/**/*counter_collection.csv", recursive=True)):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    order = []
    for row in csv.DictReader(open(f)):
        if "prior_attn" not in row["Kernel_Name"]:
            continue
        agg[row["Dispatch_Id"]][row["Counter_Name"]].append(float(row["Counter_Value"]))
    # dispatches come in groups of 6 per pattern (1 warm-up + 5 timed): average per pattern
    ids = sorted(agg, key=int)
    for pi in range(3):
        grp = ids[pi * 6:(pi + 1) * 6]
        tot = collections.defaultdict(list)
        for d in grp:
            for c, v in agg[d].items():
                tot[c].append(sum(v))
        print("pattern", pi + 1, {c: round(sum(v) / len(v)) for c, v in tot.items()})
PY
rm -rf $O
