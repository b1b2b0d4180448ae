#!/bin/bash
# Round 5, call 8: the whole -m gpu suite + smoke on the tree with predicted-statistics LayerNorm fold, lazy RoPE weights, hidden states.
mkdir -p gpurun_out/r05
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/r05/suite.log 2>&1; echo "tests exit $?"
grep -E "passed|failed" gpurun_out/r05/suite.log | tail -2; grep -E "^E  |^FAILED" gpurun_out/r05/suite.log | cut -c1-300 | head -20
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
