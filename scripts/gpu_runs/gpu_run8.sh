mkdir -p gpurun_out/pmc
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS --kernel-trace --output-format csv -d $R/gpurun_out/pmc/gemm_sq -o a -- python $R/scripts/bench_gemm.py 2 > $R/gpurun_out/pmc/gemm_sq.log 2>&1; echo "pmc 1 exit $?"
timeout 300 rocprofv3 --pmc SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA --kernel-trace --output-format csv -d $R/gpurun_out/pmc/gemm_sq2 -o a -- python $R/scripts/bench_gemm.py 2 > $R/gpurun_out/pmc/gemm_sq2.log 2>&1; echo "pmc 2 exit $?"
timeout 300 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --kernel-trace --output-format csv -d $R/gpurun_out/pmc/gemm_tcc -o a -- python $R/scripts/bench_gemm.py 2 > $R/gpurun_out/pmc/gemm_tcc.log 2>&1; echo "pmc 3 exit $?"
cd $R
python - <<'PY'
import csv, glob, collections
for d in ["gemm_sq", "gemm_sq2", "gemm_tcc"]:
    for f in glob.glob(f"gpurun_out/pmc/{d}/**/*counter_collection.csv", recursive=True):
        agg = collections.defaultdict(lambda: collections.defaultdict(list))
        for row in csv.DictReader(open(f)):
            k = row["Kernel_Name"]
            agg[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
        print("==", f)
        for k, cs in agg.items():
            if "gemm_kernel" not in k: continue
            print(k[20:75], {c: (len(v), round(sum(v)/len(v))) for c, v in cs.items()})
PY
tail -3 gpurun_out/pmc/gemm_tcc.log
