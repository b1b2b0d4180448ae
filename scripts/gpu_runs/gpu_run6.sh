mkdir -p gpurun_out/pmc
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS --kernel-trace --output-format csv -d $R/gpurun_out/pmc/attn_sq -o a -- python $R/scripts/bench_kernels.py attn > $R/gpurun_out/pmc/attn_sq.log 2>&1; echo "pmc attn sq exit $?"
timeout 300 rocprofv3 --pmc SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA --kernel-trace --output-format csv -d $R/gpurun_out/pmc/attn_sq2 -o a -- python $R/scripts/bench_kernels.py attn > $R/gpurun_out/pmc/attn_sq2.log 2>&1; echo "pmc attn sq2 exit $?"
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/pmc/gemm_fetch -o a -- python $R/scripts/bench_gemm.py 2 > $R/gpurun_out/pmc/gemm_fetch.log 2>&1; echo "pmc gemm fetch exit $?"
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/pmc/gemm_write -o a -- python $R/scripts/bench_gemm.py 2 > $R/gpurun_out/pmc/gemm_write.log 2>&1; echo "pmc gemm write exit $?"
cd $R
find gpurun_out/pmc -name "*.csv" | head -20
python - <<'PY'
import csv, glob, collections
for d in sorted(glob.glob("gpurun_out/pmc/*/")):
    files = glob.glob(d + "**/*counter_collection.csv", recursive=True)
    for f in files:
        agg = collections.defaultdict(lambda: collections.defaultdict(list))
        for row in csv.DictReader(open(f)):
            k = row["Kernel_Name"][:60]
            agg[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
        print("==", f)
        for k, cs in agg.items():
            if "llark" not in k: continue
            print(k, {c: (len(v), sum(v)/len(v)) for c, v in cs.items()})
PY
