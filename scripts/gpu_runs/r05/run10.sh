#!/bin/bash
# Round 5, call 10: B-direct kernels with the ragged last row tile issuing one live row block only (GEMM_BD_TAIL_SKIP) against the
# library as shipped: Llama-side parity tests on the new build, then the Llama stage alternating libraries (each line: split AND bf16).
mkdir -p gpurun_out/r05
out=gpurun_out/r05/run10.txt
: > $out
LLARK_HIP_LIB=$PWD/llark_amd/libllark_hip_tail.so timeout 900 python -m pytest tests/test_llama_gpu.py tests/test_prior_gpu.py -q -k "fragment or rope or engine or stream_k or two_streams or width" 2>&1 | tail -4 >> $out
LLARK_HIP_LIB=$PWD/llark_amd/libllark_hip_tail.so timeout 600 python -m pytest tests/test_fulldepth_gpu.py -q -s -k "llama and split" 2>&1 | grep -E "fulldepth\]|passed|failed" | cut -c1-300 >> $out
for rep in 1 2 3; do
for lib in libllark_hip.so libllark_hip_tail.so; do
  LLARK_HIP_LIB=$PWD/llark_amd/$lib timeout 300 python bench.py --stages llama --steps 10 --warmup 3 --no-cpu-baseline > /tmp/b.txt 2>&1
  python - $lib <<'PY' >> $out
import json, sys
ok = False
for l in open("/tmp/b.txt"):
    if l.startswith("{"):
        d = json.loads(l); ok = True
        r, o = d["roofline_llm"], d.get("roofline_llm_bf16") or {}
        print(sys.argv[1], "split ms", d["ms_per_step"], "gemm frac", r["frac"], "whole", r.get("whole_forward_frac"),
              "| bf16 ms", o.get("llama_ms_per_step"), "gemm frac", o.get("frac"), "whole", o.get("whole_forward_frac"))
if not ok:
    print(sys.argv[1], "FAILED", open("/tmp/b.txt").read()[-1500:])
PY
done; done
cat $out
