#!/bin/bash
# Round 5: memory-side traffic (FETCH_SIZE x 2 on gfx950 + WRITE_SIZE, separate --pmc passes, --kernel-trace only) of the conv stack
# (every kernel of one llark encode_top call: fused stages, codebook search with certificate, near-tie fix-up) and of the Llama stage's
# B-direct GEMMs -> profiles/r05_pmc_conv.json, profiles/r05_pmc_llm.json (bench.py's roofline_conv.traffic / roofline_llm.traffic).
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05/pmc_cl
mkdir -p $O
export TMPDIR=/tmp
cd /tmp
for st in jukebox llama; do
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 400 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/${st}_$c -o a -- python $R/bench.py --stages $st --steps 2 --warmup 1 --no-cpu-baseline --no-extras --no-alt-precision > $O/${st}_$c.log 2>&1; echo "pmc $st $c exit $?"
done; done
cd $R
python - <<'PY' | tee gpurun_out/r05/pmc_conv_llm_summary.txt
import csv, glob, collections, json
def collect(stage, counter):
    agg = collections.defaultdict(lambda: [0.0, 0])
    for f in glob.glob(f"gpurun_out/r05/pmc_cl/{stage}_{counter}/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            if row["Counter_Name"] == counter:
                a = agg[row["Kernel_Name"]]
                a[0] += float(row["Counter_Value"]); a[1] += 1
    return agg
# ---- conv stack: kernels of encode_top; calls = number of codebook_argmin_tie launches
fe, wr = collect("jukebox", "FETCH_SIZE"), collect("jukebox", "WRITE_SIZE")
conv_keys = [k for k in fe if any(s in k for s in ("vq_stage", "codebook_argmin", "resblock_mfma", "conv_mfma", "conv1d", "window", "fix", "gather_win"))]
calls = max([fe[k][1] for k in fe if "codebook_argmin_tie" in k] or [1])
rows = {k[:90]: {"launches_per_call": round(fe[k][1] / calls, 2), "read_MB_per_call_FETCHx2": round(2 * fe[k][0] * 1024 / calls / 1e6, 1), "write_MB_per_call": round(wr.get(k, [0, 0])[0] * 1024 / calls / 1e6, 1)} for k in sorted(conv_keys)}
tot_r = sum(v["read_MB_per_call_FETCHx2"] for v in rows.values()); tot_w = sum(v["write_MB_per_call"] for v in rows.values())
conv = {"source": "rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE (separate passes) over bench.py --stages jukebox --steps 2 --warmup 1 (scripts/gpu_runs/r05/pmc_conv_llm.sh); per encode_top call of 8 clips; FETCH_SIZE x 2 (gfx950 correction, MI355X_MICROARCH.md)",
        "encode_calls_profiled": calls, "kernels": rows, "traffic_bytes_per_call": int((tot_r + tot_w) * 1e6), "clips_per_call": 8,
        "algorithmic_bytes_per_call": int(8 * 1.4243e9), "traffic_over_algorithmic": round((tot_r + tot_w) * 1e6 / (8 * 1.4243e9), 3)}
json.dump(conv, open("gpurun_out/r05/pmc_conv.json", "w"), indent=1)
print(json.dumps(conv, indent=1)[:3000])
# ---- Llama stage: B-direct GEMM kernels, split flow (the stage's main flow) and bf16 (the alternate, same process)
fe, wr = collect("llama", "FETCH_SIZE"), collect("llama", "WRITE_SIZE")
out = {"source": "same passes over bench.py --stages llama --steps 2 --warmup 1: per-launch averages of every gemm_bd / gemm_bd_sk kernel (Lb1 = split flow, Lb0 = bf16 flow)", "kernels": {}}
for k in sorted(fe):
    if "gemm_bd" in k:
        n = fe[k][1]
        out["kernels"][k[:110]] = {"launches": n, "read_MB_per_launch_FETCHx2": round(2 * fe[k][0] * 1024 / n / 1e6, 1), "write_MB_per_launch": round(wr.get(k, [0, 1])[0] * 1024 / max(1, wr.get(k, [0, 1])[1]) / 1e6, 1)}
json.dump(out, open("gpurun_out/r05/pmc_llm.json", "w"), indent=1)
print(json.dumps(out, indent=1)[:3500])
PY
rm -rf $O/*/
