#!/bin/bash
# Round 6, validation on the tree as committed.  Parts (FINAL_PARTS, default all): suite = every -m gpu test + smoke; bench = the default line
# exactly as the driver runs it (python bench.py); trace = rocprofv3 --kernel-trace --stats of the e2e step.
P=${FINAL_PARTS:-suite bench trace}
R=$GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
export TMPDIR=/tmp
cd /tmp && cd $R
for part in $P; do
case $part in
suite)
  timeout 1800 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/r06/final_suite.log 2>&1; echo "tests exit $?"
  grep -E "passed|failed" gpurun_out/r06/final_suite.log | tail -2; grep -E "^E  |^FAILED" gpurun_out/r06/final_suite.log | cut -c1-300 | head -20
  timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 ;;
bench)
  timeout 1500 python bench.py > gpurun_out/r06/final_bench_default.log 2>&1; echo "bench exit $?"
  grep "^{" gpurun_out/r06/final_bench_default.log | tail -1 > gpurun_out/r06/final_bench_default.json
  python - <<'PY'
import json
d = json.loads(open("gpurun_out/r06/final_bench_default.json").read())
r = d["roofline"]
print("value", d["value"], "ms", d["ms_per_step"], "dtype", d["dtype"], "| roofline frac", r["frac"], "avg_launch_ms", r["avg_launch_ms"], "launches", r["launches"])
print("conv", {k: d["roofline_conv"].get(k) for k in ("frac", "achieved")}, "vq", d.get("vq_codes"))
print("llm", {k: d["roofline_llm"].get(k) for k in ("frac", "avg_launch_ms")}, "bf16", {k: (d.get("roofline_llm_bf16") or {}).get(k) for k in ("frac", "whole_forward_frac", "llama_ms_per_step", "parity")})
print("extra", {k: {kk: v.get(kk) for kk in ("value", "ms_per_step", "decode_ms_per_token", "mfu", "error") if isinstance(v, dict) and kk in v} for k, v in (d.get("extra") or {}).items()})
print("cpu", d["cpu_baseline"])
PY
  ;;
trace)
  ( timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/r06/prof_e2e -o a -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras --no-alt-precision > gpurun_out/r06/final_trace.log 2>&1
    f=$(find gpurun_out/r06/prof_e2e -name "*.db" | head -1); python scripts/rocprof_summary.py $f gpurun_out/r06/final_e2e_kernel_stats.txt )
  rm -rf gpurun_out/r06/prof_e2e
  head -16 gpurun_out/r06/final_e2e_kernel_stats.txt | cut -c1-170 ;;
esac
done
