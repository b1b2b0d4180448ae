#!/bin/bash
# Round 5, round-end validation on the tree as committed.  Parts (FINAL_PARTS, default all): suite = every -m gpu test + smoke;
# bench = the default line exactly as the driver runs it (python bench.py); trace = rocprofv3 --kernel-trace --stats of the e2e step;
# pmc = the PMC record of the dominant kernel (separate --pmc passes, --kernel-trace only) -> profiles/r05_pmc_gemm256x.json
P=${FINAL_PARTS:-suite bench trace pmc}
R=$GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
export TMPDIR=/tmp
cd /tmp && cd $R
for part in $P; do
case $part in
suite)
  timeout 1500 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/r05/final_suite.log 2>&1; echo "tests exit $?"
  grep -E "passed|failed" gpurun_out/r05/final_suite.log | tail -2; grep -E "^E  |^FAILED" gpurun_out/r05/final_suite.log | cut -c1-300 | head -20
  timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 ;;
bench)
  timeout 1200 python bench.py > gpurun_out/r05/final_bench_default.log 2>&1; echo "bench exit $?"
  grep "^{" gpurun_out/r05/final_bench_default.log | tail -1 > gpurun_out/r05/final_bench_default.json
  python - <<'PY'
import json
d = json.loads(open("gpurun_out/r05/final_bench_default.json").read())
r = d["roofline"]
print("value", d["value"], "ms", d["ms_per_step"], "dtype", d["dtype"], "| roofline frac", r["frac"], "avg_launch_ms", r["avg_launch_ms"], "launches", r["launches"])
print("conv", {k: d["roofline_conv"].get(k) for k in ("frac", "achieved")}, "vq", d.get("vq_codes"))
print("llm", {k: d["roofline_llm"].get(k) for k in ("frac", "avg_launch_ms")}, "bf16", {k: (d.get("roofline_llm_bf16") or {}).get(k) for k in ("frac", "whole_forward_frac", "llama_ms_per_step", "parity")})
print("extra", {k: {kk: v.get(kk) for kk in ("value", "ms_per_step", "decode_ms_per_token", "mfu") if isinstance(v, dict) and kk in v} for k, v in (d.get("extra") or {}).items()})
print("cpu", d["cpu_baseline"])
PY
  ;;
trace)
  ( timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/r05/prof_e2e -o a -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras --no-alt-precision > gpurun_out/r05/final_trace.log 2>&1
    f=$(find gpurun_out/r05/prof_e2e -name "*.db" | head -1); python scripts/rocprof_summary.py $f gpurun_out/r05/final_e2e_kernel_stats.txt )
  rm -rf gpurun_out/r05/prof_e2e
  head -16 gpurun_out/r05/final_e2e_kernel_stats.txt | cut -c1-170 ;;
pmc)
  O=$R/gpurun_out/r05/pmc_full
  mkdir -p $O
  cd /tmp
  for c in FETCH_SIZE "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum" "GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA"; do
    n=$(echo $c | cut -d' ' -f1)
    timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/$n -o a -- python $R/scripts/bench_gemm_ln.py 65536 2 > $O/$n.log 2>&1; echo "pmc $n exit $?"
  done
  cd $R
  python - <<'PY' | tee gpurun_out/r05/final_pmc_gemm256x.json
import csv, glob, collections, json
# roles of the default step: kernel name carries <dtype, EPI, LN>; Li2ELi1E = c_fc consumer (QGELU_SPLIT, LN 1), Li1ELi2E = producers (RESID, LN 2), Li0ELi1E = c_attn consumer
roles = {"gemm256x_kernelIDF16_Li2ELi1E": "c_fc consumer <f16, EPI_QGELU_SPLIT, LN 1> M=65536 N=4800 K=4800",
         "gemm256x_kernelIDF16_Li0ELi1E": "c_attn consumer <f16, EPI_F32, LN 1> M=65536 N=3648 K=4800",
         "gemm256x_kernelIDF16_Li1ELi2E": "producers <f16, EPI_RESID, LN 2> (c_proj K=1216 and c_proj2 K=4800 launches averaged)"}
out = {"source": "rocprofv3 --pmc, three separate passes (FETCH_SIZE | WRITE_SIZE TCC_HIT_sum TCC_MISS_sum | GRBM / SQ), --kernel-trace only, over scripts/bench_gemm_ln.py 65536 2 (scripts/gpu_runs/r05/final.sh pmc), csrc/gemm256x.hip as committed at the end of round 5", "roles": {}}
for key, desc in roles.items():
    vals, times = {}, []
    for f in sorted(glob.glob("gpurun_out/r05/pmc_full/**/*counter_collection.csv", recursive=True)):
        agg = collections.defaultdict(list)
        for row in csv.DictReader(open(f)):
            if key in row["Kernel_Name"]:
                agg[row["Counter_Name"]].append(float(row["Counter_Value"]))
        for c, v in agg.items():
            vals[c] = sum(v) / len(v)
    for f in sorted(glob.glob("gpurun_out/r05/pmc_full/GRBM_GUI_ACTIVE/**/*kernel_trace.csv", recursive=True)):
        for row in csv.DictReader(open(f)):
            if key in row["Kernel_Name"]:
                times.append((int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e6)
    if not times:
        continue
    ms = sum(times) / len(times)
    o = {"kernel": desc, "launches": len(times), "avg_launch_ms_profiled": round(ms, 4), "counters_per_launch": {k: round(v) for k, v in vals.items()}}
    if "FETCH_SIZE" in vals and "WRITE_SIZE" in vals:
        o["read_bytes_FETCH_SIZE_x2"] = int(2 * vals["FETCH_SIZE"] * 1024)
        o["write_bytes"] = int(vals["WRITE_SIZE"] * 1024)
        o["traffic_bytes_per_launch"] = o["read_bytes_FETCH_SIZE_x2"] + o["write_bytes"]
    if "GRBM_GUI_ACTIVE" in vals:
        o["effective_clock_ghz_profiled"] = round(vals["GRBM_GUI_ACTIVE"] / 8 / (ms * 1e-3) / 1e9, 3)
        if "SQ_VALU_MFMA_BUSY_CYCLES" in vals:
            o["mfma_busy_fraction_nominal_16cyc"] = round(vals["SQ_VALU_MFMA_BUSY_CYCLES"] / (vals["GRBM_GUI_ACTIVE"] / 8 * 1024), 3)
    if "TCC_HIT_sum" in vals:
        o["l2_hit_rate"] = round(vals["TCC_HIT_sum"] / (vals["TCC_HIT_sum"] + vals["TCC_MISS_sum"]), 4)
    out["roles"][key] = o
fc = out["roles"].get("gemm256x_kernelIDF16_Li2ELi1E")
if fc and "traffic_bytes_per_launch" in fc:
    out["kernel"] = fc["kernel"]
    out["traffic_bytes_per_launch"] = fc["traffic_bytes_per_launch"]
    out["algorithmic_bytes_per_launch"] = 65536 * 4800 * 4 + 4800 * 4800 * 2 + 65536 * 4800 * 4
    out["traffic_over_algorithmic"] = round(out["traffic_bytes_per_launch"] / out["algorithmic_bytes_per_launch"], 2)
    out["gfx950_fetch_correction"] = "FETCH_SIZE reports 1/2 of wide coalesced reads on gfx950 (MI355X_MICROARCH.md, HBM section): x2"
print(json.dumps(out, indent=1))
PY
  rm -rf $O/*/ ;;
esac
done
