R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02_11
mkdir -p $O
export TMPDIR=/tmp
cd $R
for f in s q; do LLARK_LO8_FORM=$f timeout 300 python scripts/debug_lo8_forms.py > $O/debug_$f.txt 2>&1; echo "== form $f exit $?"; grep -v amdgpu.ids $O/debug_$f.txt | cut -c1-300; done
timeout 900 python -m pytest tests/test_lo8_gpu.py -x -q -p no:cacheprovider > $O/t_lo8_s.log 2>&1; echo "lo8 tests (form s) exit $?"; grep -E "passed|failed|Error|error|assert" $O/t_lo8_s.log | cut -c1-300 | tail -6
