#!/bin/bash
# attention kernels: parity tests (incl. ALiBi + MPT trainer), then two PMC passes and a kernel trace of scripts/bench_attn.py
R=$GRAFT_REPO_ROOT
cd $R && export TMPDIR=/tmp

O=$R/gpurun_out/r03_pmc_attn
mkdir -p $O
cd /tmp
for c in "GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA" "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU"; do
  n=$(echo $c | cut -d' ' -f1)
  timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/$n -o a -- python $R/scripts/bench_attn.py > $O/$n.log 2>&1; echo "pmc $n exit $?"
done
cd $R
python - <<'PY' | tee gpurun_out/r03_pmc_attn_final_summary.txt
import csv, glob, collections
for f in sorted(glob.glob("gpurun_out/r03_pmc_attn/**/*counter_collection.csv", recursive=True)):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for row in csv.DictReader(open(f)):
        if "attn" not in row["Kernel_Name"]:
            continue
        k = row["Kernel_Name"][:60] + "|grid" + row["Grid_Size"]
        agg[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
    for k, cs in agg.items():
        print(k, {c: (len(v), round(sum(v) / len(v))) for c, v in cs.items()})
for f in sorted(glob.glob("gpurun_out/r03_pmc_attn/GRBM*/**/*kernel_trace.csv", recursive=True)):
    agg = collections.defaultdict(list)
    for row in csv.DictReader(open(f)):
        if "attn" in row["Kernel_Name"]:
            agg[row["Kernel_Name"][:60]].append((int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e3)
    for k, v in agg.items():
        print("us", k, len(v), round(sum(v) / len(v), 1))
PY
rm -rf $O
