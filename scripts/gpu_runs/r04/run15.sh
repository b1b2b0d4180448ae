#!/bin/bash
# round 4, run 15: the other bench stages on the round's last library (lines for profiles/): configs[1] alone, train at the bench default
# (4 x 512) and at the recipe micro-batch with gradient checkpointing, Llama stage in both precisions
mkdir -p gpurun_out/r04
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 600 python bench.py --stages jukebox --steps 5 --warmup 2 --no-cpu-baseline --no-alt-precision 2>/dev/null | tail -1 > gpurun_out/r04/bench_jukebox.json
timeout 600 python bench.py --stages train --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/r04/bench_train_4x512.json
timeout 600 python bench.py --stages llama --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/r04/bench_llama.json
python - <<'PY'
import json
for n in ("jukebox", "train_4x512", "llama"):
    d = json.load(open(f"gpurun_out/r04/bench_{n}.json"))
    print(n, d["value"], d["ms_per_step"], (d.get("roofline") or {}).get("frac"), d.get("mfu"), (d.get("roofline_llm_bf16") or {}).get("frac"), (d.get("roofline_conv") or {}).get("frac"))
PY
