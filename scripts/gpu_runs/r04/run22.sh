#!/bin/bash
# round 4, run 22: XCD start stagger in gemm256x (alternate libraries, -DG256X_STAGGER_NS=0 / 24000 / 48000 / 96000), LayerNorm fold on / off
mkdir -p gpurun_out/r04
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
: > gpurun_out/r04/run22_ab.txt
for rep in 1 2; do
for lib in libllark_hip.so libllark_hip_s24000.so libllark_hip_s48000.so libllark_hip_s96000.so; do
for fold in 1 0; do
  LLARK_HIP_LIB=$GRAFT_REPO_ROOT/llark_amd/$lib LLARK_PRIOR_LN_FOLD=$fold timeout 600 python bench.py --gpus 1 --steps 8 --warmup 3 --no-extras --no-cpu-baseline --stages jukebox > /tmp/b.txt 2>&1
  python - $fold $lib <<'PY' | tee -a gpurun_out/r04/run22_ab.txt
import json, sys
for l in open("/tmp/b.txt"):
    if l.startswith("{"):
        d = json.loads(l)
        print(sys.argv[2], "fold", sys.argv[1], "value", d["value"], "ms", d["ms_per_step"], "frac", d["roofline"]["frac"])
PY
done; done; done
