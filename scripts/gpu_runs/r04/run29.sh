#!/bin/bash
# round 4, run 29: the full-depth Jukebox fixtures with their printed errors, LayerNorm folded (default) and not
mkdir -p gpurun_out/r04
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for fold in 1 0; do
( LLARK_PRIOR_LN_FOLD=$fold timeout 900 python -m pytest tests/test_fulldepth_gpu.py -q -m gpu -s -k jukebox 2>&1 | grep -E "fulldepth|max.abs|err|passed|failed" | cut -c1-400 ) > gpurun_out/r04/run29_fulldepth_fold$fold.txt
echo "== fold $fold"; cat gpurun_out/r04/run29_fulldepth_fold$fold.txt
done
