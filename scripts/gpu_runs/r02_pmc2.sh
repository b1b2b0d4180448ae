# round 2, final kernel: PMC passes on the skewed lo8 GEMM (scripts/bench_gemm256.py 41) -- separate passes, --kernel-trace only
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02_pmc2
mkdir -p $O
export TMPDIR=/tmp
cd /tmp
for c in FETCH_SIZE WRITE_SIZE "GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA"; do
  n=$(echo $c | cut -d' ' -f1)
  timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/$n -o a -- python $R/scripts/bench_gemm256.py 41 > $O/$n.log 2>&1; echo "pmc $n exit $?"
done
cd $R
python - <<'PY' | tee gpurun_out/r02_pmc2/summary.txt
import csv, glob, collections
for f in sorted(glob.glob("gpurun_out/r02_pmc2/**/*counter_collection.csv", recursive=True)):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for row in csv.DictReader(open(f)):
        if "gemm256" not in row["Kernel_Name"]:
            continue
        k = row["Kernel_Name"][:60] + "|grid" + row["Grid_Size"]
        agg[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
    for k, cs in agg.items():
        print(k, {c: (len(v), round(sum(v) / len(v))) for c, v in cs.items()})
PY
rm -rf $O/*/
