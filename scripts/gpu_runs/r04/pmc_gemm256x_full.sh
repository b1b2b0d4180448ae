#!/bin/bash
# round 4: PMC passes on the prior's default GEMM tile (csrc/gemm256x.hip, variant 32) -- separate passes, --kernel-trace only (MI355X guide);
# every counter of profiles/r04_pmc_gemm256x.json comes from this ONE kernel version
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04/pmc_full
mkdir -p $O
export TMPDIR=/tmp
cd /tmp
for c in FETCH_SIZE "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum" "GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA" "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_LDS"; do
  n=$(echo $c | cut -d' ' -f1)
  timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/$n -o a -- python $R/scripts/bench_gemm256.py 32 > $O/$n.log 2>&1; echo "pmc $n exit $?"
done
cd $R
python - <<'PY' | tee gpurun_out/r04/pmc_gemm256x_full_summary.txt
import csv, glob, collections, json
KEY = "gemm256x_kernelIDF16_Li2E"          # fp16, EPI_QGELU_SPLIT: M=65536 N=4800 K=4800 (c_fc)
vals, times = {}, []
for f in sorted(glob.glob("gpurun_out/r04/pmc_full/**/*counter_collection.csv", recursive=True)):
    agg = collections.defaultdict(list)
    for row in csv.DictReader(open(f)):
        if KEY in row["Kernel_Name"]:
            agg[row["Counter_Name"]].append(float(row["Counter_Value"]))
    for c, v in agg.items():
        vals[c] = sum(v) / len(v)
for f in sorted(glob.glob("gpurun_out/r04/pmc_full/GRBM_GUI_ACTIVE/**/*kernel_trace.csv", recursive=True)):
    for row in csv.DictReader(open(f)):
        if KEY in row["Kernel_Name"]:
            times.append((int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e6)
ms = sum(times) / max(1, len(times))
out = {"kernel": "gemm256x_kernel<f16, EPI_QGELU_SPLIT> M=65536 N=4800 K=4800", "launches": len(times), "avg_launch_ms_profiled": round(ms, 4), "counters_per_launch": {k: round(v) for k, v in vals.items()}}
if "FETCH_SIZE" in vals and "WRITE_SIZE" in vals:
    out["traffic_bytes_per_launch"] = int(2 * vals["FETCH_SIZE"] * 1024 + vals["WRITE_SIZE"] * 1024)
out["algorithmic_bytes_per_launch"] = 65536 * 4800 * 4 + 4800 * 4800 * 2 + 65536 * 4800 * 4
if "GRBM_GUI_ACTIVE" in vals:
    out["effective_clock_ghz_profiled"] = round(vals["GRBM_GUI_ACTIVE"] / 8 / (ms * 1e-3) / 1e9, 3)
    if "SQ_VALU_MFMA_BUSY_CYCLES" in vals:
        out["mfma_busy_fraction_nominal_16cyc"] = round(vals["SQ_VALU_MFMA_BUSY_CYCLES"] / (vals["GRBM_GUI_ACTIVE"] / 8 * 1024), 3)
if "TCC_HIT_sum" in vals:
    out["l2_hit_rate"] = round(vals["TCC_HIT_sum"] / (vals["TCC_HIT_sum"] + vals["TCC_MISS_sum"]), 4)
if "SQ_LDS_IDX_ACTIVE" in vals and "GRBM_GUI_ACTIVE" in vals:
    out["lds_bank_conflict_cycles"] = round(vals.get("SQ_LDS_BANK_CONFLICT", 0))
print(json.dumps(out, indent=1))
PY
rm -rf $O/*/
