#!/bin/bash
# Round 6: memory-side traffic (FETCH_SIZE x 2 on gfx950, WRITE_SIZE; separate --pmc passes, --kernel-trace only) of the training products at M = 4096.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r06/pmc_train_traffic
mkdir -p $O
export TMPDIR=/tmp
cd /tmp
for c in FETCH_SIZE "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum"; do
  n=$(echo $c | cut -d' ' -f1)
  timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/$n -o a -- python $R/scripts/bench_gemm_train.py 220,102 4096 > $O/$n.log 2>&1; echo "pmc $n exit $?"
done
cd $R
python - <<'PY' | tee gpurun_out/r06/pmc_train_traffic.txt
import csv, glob, collections
vals = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob("gpurun_out/r06/pmc_train_traffic/**/*counter_collection.csv", recursive=True)):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"]
        if "gemm_bda" in k:
            vals[(k[:64], row["Grid_Size"])][row["Counter_Name"]].append(float(row["Counter_Value"]))
print("# per launch: memory-side reads (FETCH_SIZE x 2 x 1 KiB on gfx950) / writes (WRITE_SIZE x 1 KiB), L2 hit rate; M = 4096 tokens, Llama-2-7B layer shapes")
print("# algorithmic bytes: dW qkv 12288x4096 (+=): 2 x 201 MB fp32 + operands 134 MB = 537 MB; dW gate_up 22016x4096: 2 x 361 + 214 = 935 MB; dW down 4096x11008: 2 x 180 + 124 = 485 MB")
for (k, g), c in sorted(vals.items()):
    m = {n: sum(v) / len(v) for n, v in c.items()}
    line = f"{k} grid {g}: launches {len(next(iter(c.values())))}"
    if "FETCH_SIZE" in m: line += f" | reads {2 * m['FETCH_SIZE'] * 1024 / 1e6:8.1f} MB"
    if "WRITE_SIZE" in m: line += f" | writes {m['WRITE_SIZE'] * 1024 / 1e6:8.1f} MB"
    if "TCC_HIT_sum" in m: line += f" | L2 hit {m['TCC_HIT_sum'] / (m['TCC_HIT_sum'] + m['TCC_MISS_sum']):.3f}"
    print(line)
PY
rm -rf $O/*/
