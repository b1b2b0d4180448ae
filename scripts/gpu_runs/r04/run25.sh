#!/bin/bash
# round 4, run 18: last validation of the committed tree (nt weight stream in the decode Linears): whole GPU suite, smoke(), default bench line
mkdir -p gpurun_out/r04
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
( timeout 1500 python -m pytest tests -q -m gpu 2>&1 | tail -6 ) > gpurun_out/r04/run25_suite.txt
( timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 ) > gpurun_out/r04/run25_smoke.txt
( time timeout 1500 python bench.py --gpus 1 --steps 20 --warmup 5 ) > gpurun_out/r04/run25_bench.txt 2>&1
tail -2 gpurun_out/r04/run25_suite.txt; cat gpurun_out/r04/run25_smoke.txt
python - <<'PY'
import json
for l in open("gpurun_out/r04/run25_bench.txt"):
    if l.startswith("{"):
        d = json.loads(l)
        print("value", d["value"], "ms", d["ms_per_step"], "frac", d["roofline"]["frac"], "conv", d["roofline_conv"]["frac"], "vq", d["vq_codes"]["code_mismatches_vs_exact"], d["vq_codes"]["near_tie_tokens_reevaluated_exactly"], "llm", d["roofline_llm"]["frac"], d["roofline_llm_bf16"]["frac"], d["roofline_llm_bf16"]["llama_ms_per_step"])
        for k, v in d["extra"].items():
            print(k, {kk: v[kk] for kk in v if kk in ("value", "ms_per_step", "ms_per_clip", "mfu", "error", "decode_ms_per_token")}, (v.get("roofline") or {}).get("frac"))
PY
grep real gpurun_out/r04/run25_bench.txt
