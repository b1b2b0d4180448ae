#!/bin/bash
# round 4: one SQ/GRBM counter pass over variants 31 (32x32x16) and 32 (16x16x32) of the prior's GEMM tile in ONE process:
# effective clock (GRBM_GUI_ACTIVE / launch time) and matrix-pipe busy fraction of both
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04/pmc_gemm256x
mkdir -p $O
export TMPDIR=/tmp
cd /tmp
for c in "GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA" ${PMC_EXTRA}; do
  n=$(echo $c | cut -d' ' -f1)
  timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/$n -o a -- python $R/scripts/bench_gemm256.py ${PMC_VARIANTS:-31,32} > $O/$n.log 2>&1; echo "pmc $n exit $?"
done
cd $R
python - <<'PY' | tee gpurun_out/r04/pmc_gemm256x_summary.txt
import csv, glob, collections
times = collections.defaultdict(list)
for f in sorted(glob.glob("gpurun_out/r04/pmc_gemm256x/**/*kernel_trace.csv", recursive=True)):
    for row in csv.DictReader(open(f)):
        if "gemm256" in row["Kernel_Name"]:
            times[row["Kernel_Name"][:60]].append((int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e3)
for f in sorted(glob.glob("gpurun_out/r04/pmc_gemm256x/**/*counter_collection.csv", recursive=True)):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for row in csv.DictReader(open(f)):
        if "gemm256" not in row["Kernel_Name"]:
            continue
        agg[row["Kernel_Name"][:60]][row["Counter_Name"]].append(float(row["Counter_Value"]))
    for k, cs in sorted(agg.items()):
        m = {c: sum(v) / len(v) for c, v in cs.items()}
        us = sum(times[k]) / max(1, len(times[k]))
        line = f"{k}: launches {len(times[k])} avg {us:.1f} us"
        if "GRBM_GUI_ACTIVE" in m:
            line += f" | clock {m['GRBM_GUI_ACTIVE'] / us / 1e3:.3f} GHz"
        if "SQ_VALU_MFMA_BUSY_CYCLES" in m and "GRBM_GUI_ACTIVE" in m:
            line += f" | MFMA busy {m['SQ_VALU_MFMA_BUSY_CYCLES'] / (m['GRBM_GUI_ACTIVE'] * 1024):.3f} of SIMD-cycles, insts {m.get('SQ_INSTS_MFMA', 0):.0f}"
        if "SQ_WAVE_CYCLES" in m:
            line += f" | wait_any {m.get('SQ_WAIT_ANY', 0) / m['SQ_WAVE_CYCLES']:.3f} wait_inst {m.get('SQ_WAIT_INST_ANY', 0) / m['SQ_WAVE_CYCLES']:.3f} active {m.get('SQ_ACTIVE_INST_ANY', 0) / m['SQ_WAVE_CYCLES']:.3f}"
        print(line)
PY
rm -rf $O/*/
