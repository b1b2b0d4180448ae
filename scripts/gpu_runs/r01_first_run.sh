mkdir -p gpurun_out
export TMPDIR=/tmp
rocminfo | grep -E "Marketing Name|Compute Unit" | head -4 > gpurun_out/hw.txt; nproc >> gpurun_out/hw.txt; free -g | head -2 >> gpurun_out/hw.txt
for f in test_vqvae_gpu test_prior_gpu test_extract_gpu; do
  timeout 600 python -m pytest tests/$f.py -m gpu -q --tb=short -rA -p no:cacheprovider > gpurun_out/$f.log 2>&1; echo "$f exit $?" >> gpurun_out/summary.txt
done
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; echo "smoke exit $?" >> gpurun_out/summary.txt
timeout 900 python bench.py --stages jukebox --steps 2 --warmup 1 > gpurun_out/bench_jukebox.log 2>&1; echo "bench exit $?" >> gpurun_out/summary.txt
cat gpurun_out/summary.txt; tail -5 gpurun_out/bench_jukebox.log
