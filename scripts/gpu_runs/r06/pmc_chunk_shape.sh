#!/bin/bash
# Round 6: gemm256x tile order, tile rows per M-group GM = 2 / 4 (default) / 8 / 16 (a chunk of 32 resident workgroups per
# XCD = GM x 32/GM tiles): un-profiled time of the four roles, then FETCH_SIZE and GRBM / SQ counter passes per variant.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r06/pmc_chunk
mkdir -p $O
export TMPDIR=/tmp
cd /tmp
for g in 4 2 8 16; do
  lib=$R/llark_amd/libllark_hip.so; [ $g != 4 ] && lib=$R/llark_amd/libllark_hip_gm$g.so
  echo "== GM=$g ($lib)"
  LLARK_HIP_LIB=$lib timeout 300 python $R/scripts/bench_gemm_ln.py 65536 3 2>&1 | grep -E "consumer|producer" | cut -c1-150
  for c in FETCH_SIZE "GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY TCC_HIT_sum TCC_MISS_sum"; do
    n=$(echo $c | cut -d' ' -f1)
    LLARK_HIP_LIB=$lib timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/gm$g/$n -o a -- python $R/scripts/bench_gemm_ln.py 65536 1 > $O/gm${g}_$n.log 2>&1; echo "pmc GM=$g $n exit $?"
  done
done
cd $R
python - <<'PY' | tee gpurun_out/r06/gemm256x_chunk_shape.txt
import csv, glob, collections
roles = {"gemm256x_kernelIDF16_Li2ELi1E": "c_fc consumer", "gemm256x_kernelIDF16_Li0ELi1E": "c_attn consumer", "gemm256x_kernelIDF16_Li1ELi2E": "producers (K=1216 + K=4800)"}
print("# gemm256x tile order: GM tile rows per M-group (chunk of 32 workgroups per XCD = GM x 32/GM tiles), M = 65536, profiled passes")
for g in (2, 4, 8, 16):
    for key, desc in roles.items():
        vals, times = {}, []
        for f in sorted(glob.glob(f"gpurun_out/r06/pmc_chunk/gm{g}/**/*counter_collection.csv", recursive=True)):
            agg = collections.defaultdict(list)
            for row in csv.DictReader(open(f)):
                if key in row["Kernel_Name"]:
                    agg[row["Counter_Name"]].append(float(row["Counter_Value"]))
            for c, v in agg.items():
                vals[c] = sum(v) / len(v)
        for f in sorted(glob.glob(f"gpurun_out/r06/pmc_chunk/gm{g}/GRBM_GUI_ACTIVE/**/*kernel_trace.csv", recursive=True)):
            for row in csv.DictReader(open(f)):
                if key in row["Kernel_Name"]:
                    times.append((int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e6)
        if not times:
            continue
        ms = sum(times) / len(times)
        line = f"GM={g:2d} {desc:30s} {ms:7.3f} ms (profiled, {len(times)} launches)"
        if "FETCH_SIZE" in vals:
            line += f" | reads {2 * vals['FETCH_SIZE'] * 1024 / 1e9:6.2f} GB (FETCH_SIZE x 2)"
        if "GRBM_GUI_ACTIVE" in vals:
            line += f" | clock {vals['GRBM_GUI_ACTIVE'] / 8 / (ms * 1e-3) / 1e9:5.3f} GHz"
            if "SQ_VALU_MFMA_BUSY_CYCLES" in vals:
                line += f" | pipe busy {vals['SQ_VALU_MFMA_BUSY_CYCLES'] / (vals['GRBM_GUI_ACTIVE'] / 8 * 1024):5.3f}"
        if "TCC_HIT_sum" in vals:
            line += f" | L2 hit {vals['TCC_HIT_sum'] / (vals['TCC_HIT_sum'] + vals['TCC_MISS_sum']):5.3f}"
        print(line)
PY
rm -rf $O/gm*/
