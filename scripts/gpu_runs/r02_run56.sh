R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02_56
mkdir -p $O
export TMPDIR=/tmp
cd $R
echo "== default"; timeout 200 python scripts/bench_decode.py split 2>&1 | grep "no per-op" | grep "B=1"
echo "== LLARK_DECODE_FUSE_NORM=1"; LLARK_DECODE_FUSE_NORM=1 timeout 200 python scripts/bench_decode.py split 2>&1 | grep "no per-op" | grep "B=1"
LLARK_DECODE_FUSE_NORM=1 timeout 400 python -m pytest tests/test_llama_gpu.py -x -q -p no:cacheprovider 2>&1 | tail -2
