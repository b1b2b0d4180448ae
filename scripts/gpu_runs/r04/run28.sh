#!/bin/bash
# round 4, run 28: codebook argmin with 16 waves per 64 tokens (exactness tests, conv roofline), re-check of the rebuilt gemm256x (alignment condition of the wide stores), smoke
mkdir -p gpurun_out/r04
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
( timeout 900 python -m pytest tests/test_vqvae_gpu.py tests/test_prior_gpu.py -q -m gpu -x -k "vqvae or ln_folded or folded_layernorm or gemm256x" 2>&1 | tail -5 ) > gpurun_out/r04/run28_tests.txt
tail -3 gpurun_out/r04/run28_tests.txt
( timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 )
for rep in 1 2; do
timeout 600 python bench.py --gpus 1 --steps 10 --warmup 3 --no-extras --no-cpu-baseline --no-alt-precision > gpurun_out/r04/run28_bench.txt 2>&1
python - <<'PY'
import json
for l in open("gpurun_out/r04/run28_bench.txt"):
    if l.startswith("{"):
        d = json.loads(l)
        print("value", d["value"], "ms", d["ms_per_step"], "frac", d["roofline"]["frac"], "conv", d["roofline_conv"]["frac"], d["roofline_conv"]["ms_per_clip"], "vq", d["vq_codes"])
PY
done
