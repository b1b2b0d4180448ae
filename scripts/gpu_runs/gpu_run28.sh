mkdir -p gpurun_out/pmc4
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
for c in "TCC_HIT_sum TCC_MISS_sum" FETCH_SIZE; do
  n=$(echo $c | cut -d' ' -f1)
  LLARK_SKIP_CHECK=1 timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $R/gpurun_out/pmc4/$n -o a -- python $R/scripts/bench_gemm.py 20 > $R/gpurun_out/pmc4/$n.log 2>&1; echo "pmc $n exit $?"
done
cd $R
python - <<'PY'
import csv, glob, collections
for f in sorted(glob.glob("gpurun_out/pmc4/**/*counter_collection.csv", recursive=True)):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"] + "|grid" + row["Grid_Size"]
        agg[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
    for k, cs in agg.items():
        if "gemm_" not in k: continue
        print(k[14:70], k[-12:], {c: (len(v), round(sum(v)/len(v))) for c, v in cs.items()})
PY
