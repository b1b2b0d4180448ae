mkdir -p gpurun_out/pmc5
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
for c in "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_BUSY_CYCLES"; do
  n=$(echo $c | cut -d' ' -f1)
  timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $R/gpurun_out/pmc5/vq_$n -o a -- python $R/scripts/bench_kernels.py vqvae > $R/gpurun_out/pmc5/vq_$n.log 2>&1; echo "pmc vq $n exit $?"
done
timeout 600 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $R/gpurun_out/pmc5/llama_sq -o a -- python $R/bench.py --stages llama --llm-precision bf16 --steps 2 --warmup 1 --no-cpu-baseline > $R/gpurun_out/pmc5/llama_sq.log 2>&1; echo "pmc llama exit $?"
cd $R
python - <<'PY'
import csv, glob, collections
for d in sorted(glob.glob("gpurun_out/pmc5/*/")):
    for f in glob.glob(d + "**/*counter_collection.csv", recursive=True):
        agg = collections.defaultdict(lambda: collections.defaultdict(list))
        wall = collections.defaultdict(list)
        for row in csv.DictReader(open(f)):
            k = row["Kernel_Name"]
            agg[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
            wall[k].append(int(row["End_Timestamp"]) - int(row["Start_Timestamp"]))
        print("==", d)
        tot = collections.defaultdict(float); tw = 0
        for k, cs in agg.items():
            if "llark" not in k: continue
            if not any(s in k for s in ("resblock", "conv_mfma", "conv1d", "codebook", "gemm_", "attn_prefill")): continue
            n = len(wall[k]); w = sum(wall[k])
            m = {c: sum(v) for c, v in cs.items()}
            line = f"{k[10:60]:52s} calls {n:5d} wall_ms {w/1e6:8.3f} " + " ".join(f"{c}={v:.4g}" for c, v in m.items())
            if "GRBM_GUI_ACTIVE" in m and w:
                line += f" | clk {m['GRBM_GUI_ACTIVE']/8/w:.2f} GHz mfma_busy {m.get('SQ_VALU_MFMA_BUSY_CYCLES',0)/1024/(m['GRBM_GUI_ACTIVE']/8):.3f}"
            print(line)
PY
