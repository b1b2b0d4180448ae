mkdir -p gpurun_out/pmc3
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_prior_gpu.py -m gpu -q --tb=short -x -p no:cacheprovider -k "gemm" > gpurun_out/tests26.log 2>&1; echo "tests exit $?"
grep -E "passed|failed" gpurun_out/tests26.log | tail -2; grep -E "^E  " gpurun_out/tests26.log | cut -c1-300 | head -20
cd /tmp
for c in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum"; do
  n=$(echo $c | cut -d' ' -f1)
  LLARK_SKIP_CHECK=1 timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $R/gpurun_out/pmc3/$n -o a -- python $R/scripts/bench_gemm.py 12 > $R/gpurun_out/pmc3/$n.log 2>&1; echo "pmc $n exit $?"
done
cd $R
python - <<'PY'
import csv, glob, collections
for f in sorted(glob.glob("gpurun_out/pmc3/**/*counter_collection.csv", recursive=True)):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"] + "|grid" + row["Grid_Size"]
        agg[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
    for k, cs in agg.items():
        if "gemm_kernel" not in k: continue
        print(k[14:70], k[-12:], {c: (len(v), round(sum(v)/len(v))) for c, v in cs.items()})
PY
timeout 600 python bench.py --steps 3 --warmup 1 > gpurun_out/bench_e2e.log 2>&1; echo "bench exit $?"; tail -c 3400 gpurun_out/bench_e2e.log
