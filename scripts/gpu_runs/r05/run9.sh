#!/bin/bash
# Round 5, call 9: gemm256x epilogue with staggered residual loads (load 0, 1 | store 0 | load 2 | store 1 | load 3 | store 2 | store 3) and
# the two-FMA folded-LayerNorm consumer, against the library as shipped (two halves): parity tests on the new build, role microbenchmark
# + jukebox stage alternating libraries (two repetitions), per-tile stamps of the new build.
mkdir -p gpurun_out/r05
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/r05/run9.txt
: > $out
LLARK_HIP_LIB=$PWD/llark_amd/libllark_hip_epi2.so timeout 600 python -m pytest tests/test_prior_gpu.py -q -k "ln or gemm256x or prior_full or prior_folded or epilogues" 2>&1 | tail -4 >> $out
for rep in 1 2; do
for lib in libllark_hip.so libllark_hip_epi2.so; do
  echo "== $lib rep $rep" >> $out
  LLARK_HIP_LIB=$PWD/llark_amd/$lib timeout 300 python scripts/bench_gemm_ln.py 65536 5 2>&1 | grep -v "amdgpu.ids\|^{" >> $out
  LLARK_HIP_LIB=$PWD/llark_amd/$lib timeout 600 python bench.py --gpus 1 --steps 8 --warmup 3 --no-extras --no-cpu-baseline --stages jukebox > /tmp/b.txt 2>&1
  python - $lib <<'PY' >> $out
import json, sys
for l in open("/tmp/b.txt"):
    if l.startswith("{"):
        d = json.loads(l)
        print(sys.argv[1], "jukebox stage: value", d["value"], "ms", d["ms_per_step"], "frac", d["roofline"]["frac"], "kernel_ms", {k: v for k, v in d["kernel_ms"].items() if "gemm" in k})
PY
done; done
if [ -f llark_amd/libllark_hip_epi2prof.so ]; then
  echo "== per-tile stamps, staggered-load build" >> $out
  LLARK_HIP_LIB=$PWD/llark_amd/libllark_hip_epi2prof.so timeout 300 python scripts/probes/gemm256x_tile_times.py 256 2>&1 | grep -v amdgpu.ids >> $out
fi
cat $out
