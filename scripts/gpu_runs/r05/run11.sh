#!/bin/bash
# Round 5, call 11: gemm_bda (A by LDS-DMA) as the default of the headline Llama flow, per-tile AND K-cut kernels: every Llama-side and
# fragment-major test, the 7B full-depth fixture, smoke, the kernel microbenchmark and the Llama stage.
mkdir -p gpurun_out/r05
out=gpurun_out/r05/run11.txt
: > $out
timeout 900 python -m pytest tests/test_llama_gpu.py tests/test_prior_gpu.py tests/test_infer_driver.py tests/test_gemv_dma_gpu.py -q -k "not ln and not layernorm and not attention" 2>&1 | tail -4 >> $out
timeout 600 python -m pytest tests/test_fulldepth_gpu.py -q -s -k "llama" 2>&1 | grep -E "fulldepth\]|passed|failed" | cut -c1-260 >> $out
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 >> $out
python scripts/bench_gemm_bda.py 2968 5 2>&1 | grep -v amdgpu.ids >> $out
for rep in 1 2; do
  timeout 300 python bench.py --stages llama --steps 10 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); r=d['roofline_llm']; o=d['roofline_llm_bf16']
print('llama stage: split ms', d['ms_per_step'], 'gemm frac', r['frac'], 'whole', r.get('whole_forward_frac'), '| bf16 ms', o['llama_ms_per_step'], 'whole', o['whole_forward_frac'], 'parity', o['parity']['diff_over_max'])" >> $out
done
cat $out
