#!/bin/bash
# round 4: memory-side bytes of the folded-LayerNorm roles against the plain products (separate --pmc passes as the guide's HBM
# section prescribes; the summary applies the same gfx950 correction as profiles/r04_pmc_gemm256x.json: FETCH_SIZE x2)
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04/pmc_gemm_ln
mkdir -p $O
export TMPDIR=/tmp
cd /tmp
timeout 300 python $R/scripts/bench_gemm_ln.py 65536 5 > $R/gpurun_out/r04/bench_gemm_ln.txt 2>&1
cat $R/gpurun_out/r04/bench_gemm_ln.txt | head -14
for c in "FETCH_SIZE" "WRITE_SIZE"; do
  n=$(echo $c | cut -d' ' -f1)
  timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/$n -o a -- python $R/scripts/bench_gemm_ln.py 65536 1 > $O/$n.log 2>&1; echo "pmc $n exit $?"
done
cd $R
python - <<'PY' | tee gpurun_out/r04/pmc_gemm_ln_summary.txt
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob("gpurun_out/r04/pmc_gemm_ln/**/*counter_collection.csv", recursive=True)):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"]
        if "gemm256" not in k and "layernorm" not in k and "ln_stats" not in k:
            continue
        agg[k[:70] + " grid " + row.get("Grid_Size", "")][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, cs in sorted(agg.items()):
    m = {c: sum(v) / len(v) for c, v in cs.items()}
    # FETCH_SIZE / WRITE_SIZE are in KB; FETCH_SIZE x2 on gfx950 for wide coalesced reads (MI355X_MICROARCH.md, HBM section)
    print(f"{k}: launches {len(next(iter(cs.values())))}  read {2 * m.get('FETCH_SIZE', 0) * 1024 / 1e9:.3f} GB (FETCH_SIZE x2)  write {m.get('WRITE_SIZE', 0) * 1024 / 1e9:.3f} GB")
PY
rm -rf $O/*/
