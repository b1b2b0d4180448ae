#!/bin/bash
# round 4, run 26: Llama prefill with the LDS-staged persistent tiles (LLARK_FRAG=0 -> pick_variant, gemm256x where >= 384 tiles) against the B-direct default
mkdir -p gpurun_out/r04
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
: > gpurun_out/r04/run26_llama_frag_ab.txt
for rep in 1 2; do for frag in 1 0; do
  LLARK_FRAG=$frag timeout 300 python bench.py --gpus 1 --steps 10 --warmup 3 --stages llama --no-extras --no-cpu-baseline > /tmp/b.txt 2>&1
  python - $frag <<'PY' | tee -a gpurun_out/r04/run26_llama_frag_ab.txt
import json, sys
for l in open("/tmp/b.txt"):
    if l.startswith("{"):
        d = json.loads(l)
        r = d.get("roofline_llm") or d.get("roofline") or {}
        print("LLARK_FRAG", sys.argv[1], "ms", d["ms_per_step"], "value", d["value"], "frac", r.get("frac"), "bf16", (d.get("roofline_llm_bf16") or {}).get("frac"), (d.get("roofline_llm_bf16") or {}).get("llama_ms_per_step"))
PY
done; done
tail -3 /tmp/b.txt | cut -c1-300
