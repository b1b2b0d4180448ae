R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02_12
mkdir -p $O
export TMPDIR=/tmp
cd $R
LLARK_LO8_FORM=q timeout 300 python scripts/debug_lo8_forms.py > $O/debug_q.txt 2>&1; echo "== form q exit $?"; grep -v amdgpu.ids $O/debug_q.txt | cut -c1-300
