mkdir -p gpurun_out/pmc2
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
for ab in 0 11 12 14; do
  if [ $ab != 0 ]; then export LLARK_HIP_LIB=$R/llark_amd/libllark_hip_ab$ab.so; fi
  LLARK_SKIP_CHECK=1 timeout 300 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA --kernel-trace --output-format csv -d $R/gpurun_out/pmc2/ab$ab -o a -- python $R/scripts/bench_gemm.py 100 > $R/gpurun_out/pmc2/ab$ab.log 2>&1; echo "pmc ab$ab exit $?"
done
cd $R
python - <<'PY'
import csv, glob, collections
for d in ["ab0", "ab11", "ab12", "ab14"]:
    for f in glob.glob(f"gpurun_out/pmc2/{d}/**/*counter_collection.csv", recursive=True):
        agg = collections.defaultdict(lambda: collections.defaultdict(list))
        wall = collections.defaultdict(list)
        for row in csv.DictReader(open(f)):
            k = row["Kernel_Name"] + "|grid" + row["Grid_Size"]
            agg[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
            wall[k].append(int(row["End_Timestamp"]) - int(row["Start_Timestamp"]))
        print("==", d)
        for k, cs in agg.items():
            if "gemm_bd" not in k: continue
            w = sum(wall[k]) / len(wall[k])
            m = {c: sum(v)/len(v) for c, v in cs.items()}
            print(k[14:60], k[-12:], "wall_us %.0f" % (w/1e3), {c: round(v) for c, v in m.items()}, "clk_GHz %.2f" % (m.get("GRBM_GUI_ACTIVE", 0) / w) )
PY
