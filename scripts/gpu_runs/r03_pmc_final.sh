#!/bin/bash
# round 3: PMC passes on the FINAL two-pass 256x256 tile (csrc/gemm256n.hip, G256N_SCHED = 3) -- separate passes, --kernel-trace only;
# every counter of profiles/r03_pmc_gemm256n.json comes from this ONE kernel version
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03_pmc_final
mkdir -p $O
export TMPDIR=/tmp
cd /tmp
for c in FETCH_SIZE "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum" "GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA" "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_LDS"; do
  n=$(echo $c | cut -d' ' -f1)
  timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/$n -o a -- python $R/scripts/bench_gemm256.py 31 > $O/$n.log 2>&1; echo "pmc $n exit $?"
done
cd $R
python - <<'PY' | tee gpurun_out/r03_pmc_final/summary.txt
import csv, glob, collections
for f in sorted(glob.glob("gpurun_out/r03_pmc_final/**/*counter_collection.csv", recursive=True)):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for row in csv.DictReader(open(f)):
        if "gemm256" not in row["Kernel_Name"]:
            continue
        k = row["Kernel_Name"][:70] + "|grid" + row["Grid_Size"]
        agg[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
    for k, cs in agg.items():
        print(k, {c: (len(v), round(sum(v) / len(v))) for c, v in cs.items()})
PY
python - <<'PY' | tee gpurun_out/r03_pmc_final/kernel_times.txt
import csv, glob, collections
for f in sorted(glob.glob("gpurun_out/r03_pmc_final/*/**/*kernel_trace.csv", recursive=True)):
    agg = collections.defaultdict(list)
    for row in csv.DictReader(open(f)):
        if "gemm256" in row["Kernel_Name"]:
            agg[row["Kernel_Name"][:70]].append((int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e6)
    print(f.split("/")[2], {k[-30:]: (len(v), round(sum(v) / len(v), 4)) for k, v in agg.items()})
PY
rm -rf $O/*/
