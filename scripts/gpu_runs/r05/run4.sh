#!/bin/bash
# Round 5, call 4: Llama prefill as two half-batches on two streams (LLARK_PREFILL_STREAMS=2) against one stream: parity test, then the
# Llama stage of bench.py in both settings, alternating, two repetitions (each line carries split AND bf16).
mkdir -p gpurun_out/r05
out=gpurun_out/r05/run4.txt
: > $out
timeout 300 python -m pytest tests/test_llama_gpu.py -q -k "two_streams" 2>&1 | tail -6 >> $out
for rep in 1 2; do
for st in 1 2; do
  LLARK_PREFILL_STREAMS=$st timeout 300 python bench.py --stages llama --steps 10 --warmup 3 --no-cpu-baseline > /tmp/b.txt 2>&1
  python - $st <<'PY' >> $out
import json, sys
ok = False
for l in open("/tmp/b.txt"):
    if l.startswith("{"):
        d = json.loads(l); ok = True
        r, o = d["roofline_llm"], d.get("roofline_llm_bf16") or {}
        print("streams", sys.argv[1], "split ms", d["ms_per_step"], "gemm frac", r["frac"], "whole", r.get("whole_forward_frac"),
              "| bf16 ms", o.get("llama_ms_per_step"), "gemm frac", o.get("frac"), "whole", o.get("whole_forward_frac"), "parity", (o.get("parity") or {}).get("diff_over_max"), (o.get("parity") or {}).get("argmax_agree_frac"))
if not ok:
    print("streams", sys.argv[1], "FAILED", open("/tmp/b.txt").read()[-1500:])
PY
done; done
cat $out
