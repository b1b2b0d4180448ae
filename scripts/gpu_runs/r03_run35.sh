#!/bin/bash
cd $GRAFT_REPO_ROOT && export TMPDIR=/tmp
for st in "train --batch 8 --micro-batch 2 --train-seq 2048" "train --batch 8 --micro-batch 2 --train-seq 2048 --grad-checkpoint" "mpt-train" "mpt" "clap" "jukebox"; do
  n=$(echo $st | tr ' -' '__')
  timeout 900 python bench.py --stages $st --no-cpu-baseline 2>&1 | tail -1 > gpurun_out/r03_final_$n.json
  python -c "
import json,sys
try:
    d=json.load(open('gpurun_out/r03_final_$n.json')); print('$st:', d['ms_per_step'], d['value'], d['unit'], 'mfu', d.get('mfu'), 'peak', d.get('peak_hbm_gb'))
except Exception as e: print('$st: FAILED', e); print(open('gpurun_out/r03_final_$n.json').read()[-600:])"
done
