for v in "" _d1 _d2; do echo "== lib$v"; LLARK_HIP_LIB=$PWD/llark_amd/libllark_hip$v.so python scripts/probes/bda_cproj_shape.py 2>&1 | grep "PRODUCER DMA\|K=1216 bda"; done
