#!/bin/bash
# Round 5, call 3: gemm256x chunk barrier with one tile of run-ahead (and a seeded start stagger) against the strict barrier --
# role microbenchmark + the jukebox stage, alternating libraries, two repetitions; per-tile stamps of the run-ahead build.
mkdir -p gpurun_out/r05
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/r05/run3.txt
: > $out
for rep in 1 2; do
for lib in libllark_hip.so libllark_hip_ra1.so libllark_hip_ra1s.so; do
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
if [ -f llark_amd/libllark_hip_ra1prof.so ]; then
  echo "== per-tile stamps, run-ahead build" >> $out
  LLARK_HIP_LIB=$PWD/llark_amd/libllark_hip_ra1prof.so timeout 300 python scripts/probes/gemm256x_tile_times.py 256 2>&1 | grep -v amdgpu.ids >> $out
fi
cat $out
