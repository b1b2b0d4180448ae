R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02_16
mkdir -p $O
export TMPDIR=/tmp
cd $R
timeout 900 python -m pytest tests/test_lo8_gpu.py tests/test_infer_driver.py tests/test_extract_gpu.py "tests/test_llama_gpu.py::test_wrapped_model_api_loss_generate_errors" tests/test_train_gpu.py -x -q -p no:cacheprovider > $O/t_sel.log 2>&1; echo "selected tests exit $?"; grep -E "passed|failed|Error|error|assert" $O/t_sel.log | cut -c1-300 | tail -6
timeout 600 python bench.py --stages jukebox --no-cpu-baseline > $O/bench_jukebox.log 2>&1; echo "jukebox exit $?"; tail -1 $O/bench_jukebox.log | cut -c1-300
bash scripts/gpu_runs/r02_pmc.sh 2>&1 | cut -c1-400
