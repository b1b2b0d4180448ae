#!/bin/bash
# round 4, run 13: final validation of the committed tree -- whole GPU suite + smoke()
mkdir -p gpurun_out/r04
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
( timeout 1500 python -m pytest tests -q -m gpu 2>&1 | tail -8 ) > gpurun_out/r04/run13_suite.txt
( timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 ) > gpurun_out/r04/run13_smoke.txt
tail -3 gpurun_out/r04/run13_suite.txt; cat gpurun_out/r04/run13_smoke.txt
