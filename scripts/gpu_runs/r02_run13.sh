R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02_13
mkdir -p $O
export TMPDIR=/tmp
cd $R
for m in 1 2; do LLARK_HIP_LIB=$R/llark_amd/libllark_hip_qepi$m.so LLARK_LO8_FORM=q timeout 300 python scripts/debug_lo8_forms.py > $O/debug_q$m.txt 2>&1; echo "== epi mode $m exit $?"; grep -v amdgpu.ids $O/debug_q$m.txt | grep -v "hist\|tile (row\|outside" | cut -c1-250; done
