R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02_20
mkdir -p $O
export TMPDIR=/tmp
cd $R
timeout 900 python -m pytest tests/test_train_gpu.py -x -q -p no:cacheprovider > $O/t_train.log 2>&1; echo "train tests exit $?"; tail -3 $O/t_train.log | cut -c1-300
timeout 900 python bench.py --stages train --no-cpu-baseline --grad-checkpoint > $O/bench_train_ckpt.log 2>&1; echo "train ckpt exit $?: $(grep -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*\|"peak_hbm_gb": [0-9.]*' $O/bench_train_ckpt.log | tr '\n' ' ')"
timeout 900 python bench.py --stages train --no-cpu-baseline --batch 8 --micro-batch 2 --train-seq 2048 > $O/bench_train_2x2048.log 2>&1; echo "train 2x2048x4 exit $?: $(grep -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*\|"peak_hbm_gb": [0-9.]*' $O/bench_train_2x2048.log | tr '\n' ' ')"; tail -3 $O/bench_train_2x2048.log | cut -c1-300
timeout 900 python bench.py --stages train --no-cpu-baseline --batch 8 --micro-batch 2 --train-seq 2048 --grad-checkpoint > $O/bench_train_2x2048_ckpt.log 2>&1; echo "train 2x2048x4 ckpt exit $?: $(grep -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*\|"peak_hbm_gb": [0-9.]*' $O/bench_train_2x2048_ckpt.log | tr '\n' ' ')"
bash scripts/gpu_runs/r02_pmc.sh 2>&1 | grep "lo8n_kernel<7>\|exit" | cut -c1-600
