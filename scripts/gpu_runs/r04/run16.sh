#!/bin/bash
# round 4, run 16: decode weight stream with nt (aux = 2) LDS-DMA loads (libllark_hip_nt.so) against the default policy, alternating
mkdir -p gpurun_out/r04
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -f gpurun_out/r04/decode_nt_weights.txt
( timeout 300 env LLARK_HIP_LIB=$GRAFT_REPO_ROOT/llark_amd/libllark_hip_nt.so python -m pytest tests/test_gemv_dma_gpu.py -x -q 2>&1 | tail -2 ) >> gpurun_out/r04/decode_nt_weights.txt
for rep in 1 2; do
  for lib in default nt; do
    if [ $lib = default ]; then unset LLARK_HIP_LIB; else export LLARK_HIP_LIB=$GRAFT_REPO_ROOT/llark_amd/libllark_hip_nt.so; fi
    timeout 600 python bench.py --stages generate --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); r=d.get('roofline_decode') or {}; print('$lib rep $rep', 'ms/clip', d['ms_per_step'], 'decode ms/token', r.get('decode_ms_per_token'), 'skinny GB/s', r.get('achieved'), {k:v for k,v in d['kernel_ms'].items() if 'skinny' in k})
" >> gpurun_out/r04/decode_nt_weights.txt
  done
done
cat gpurun_out/r04/decode_nt_weights.txt
