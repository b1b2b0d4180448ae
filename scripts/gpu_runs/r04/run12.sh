#!/bin/bash
# round 4, run 12: the default bench line with every leg (short main loop) -- checks the new configs[4] legs and the wall time
mkdir -p gpurun_out/r04
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
( time timeout 1500 python bench.py --steps 5 --warmup 2 ) > gpurun_out/r04/run12_bench.txt 2>&1
python - <<'PY'
import json
for l in open("gpurun_out/r04/run12_bench.txt"):
    if l.startswith("{"):
        d = json.loads(l)
        print("value", d["value"], "ms", d["ms_per_step"], "conv", d["roofline_conv"]["frac"], "vq", d["vq_codes"]["code_mismatches_vs_exact"], d["vq_codes"]["without_certificate"])
        for k, v in d["extra"].items():
            print(k, {kk: v[kk] for kk in v if kk in ("value", "ms_per_step", "ms_per_clip", "mfu", "error", "decode_ms_per_token")}, (v.get("roofline") or {}).get("frac"))
PY
tail -4 gpurun_out/r04/run12_bench.txt | grep real
