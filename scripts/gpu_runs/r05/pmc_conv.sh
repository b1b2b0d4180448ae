#!/bin/bash
# Round 5: memory-side traffic of the conv stack (FETCH_SIZE x 2 on gfx950, WRITE_SIZE; separate --pmc passes, --kernel-trace only) over
# scripts/probes/vq_encode_loop.py -> profiles/r05_pmc_conv.json (bench.py's roofline_conv.traffic)
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05/pmc_conv
mkdir -p $O
export TMPDIR=/tmp
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 400 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/$c -o a -- python $R/scripts/probes/vq_encode_loop.py 5 > $O/$c.log 2>&1; echo "pmc $c exit $?"; tail -1 $O/$c.log
done
cd $R
python - <<'PY' | tee gpurun_out/r05/pmc_conv.json
import csv, glob, collections, json
def collect(counter):
    rows = []
    for f in glob.glob(f"gpurun_out/r05/pmc_conv/{counter}/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            if row["Counter_Name"] == counter:
                rows.append((int(row["Dispatch_Id"]), row["Kernel_Name"], float(row["Counter_Value"])))
    rows.sort()
    marks = [i for i, r in enumerate(rows) if "split16_v4_kernel" in r[1]]
    assert len(marks) >= 2, marks
    return rows[marks[-2] + 1: marks[-1]]
CALLS = 5
fe, wr = collect("FETCH_SIZE"), collect("WRITE_SIZE")
agg = collections.defaultdict(lambda: [0, 0.0, 0.0])
for _, k, v in fe:
    agg[k][0] += 1; agg[k][1] += v
for _, k, v in wr:
    agg[k][2] += v
kern = {k[:100]: {"launches_per_call": round(a[0] / CALLS, 1), "read_MB_per_call_FETCHx2": round(2 * a[1] * 1024 / CALLS / 1e6, 1), "write_MB_per_call": round(a[2] * 1024 / CALLS / 1e6, 1)} for k, a in sorted(agg.items())}
rd = sum(v["read_MB_per_call_FETCHx2"] for v in kern.values()) * 1e6
wrb = sum(v["write_MB_per_call"] for v in kern.values()) * 1e6
alg = 8 * 1.4243e9
print(json.dumps({"source": "rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE (two separate passes, --kernel-trace only) over scripts/probes/vq_encode_loop.py 5 (scripts/gpu_runs/r05/pmc_conv.sh): every kernel launched by 5 encode_top calls of the default encoder on 8 clips, cut out between two marker launches; per call",
                  "kernel": "encode_top (7 vq_stage launches + codebook search with certificate + near-tie fix-up on the exact kernels), 8 clips", "clips_per_call": 8,
                  "kernels": kern, "read_bytes_FETCH_SIZE_x2": int(rd), "write_bytes": int(wrb), "traffic_bytes_per_call": int(rd + wrb), "algorithmic_bytes_per_call": int(alg),
                  "traffic_over_algorithmic": round((rd + wrb) / alg, 3), "gfx950_fetch_correction": "FETCH_SIZE reports 1/2 of wide coalesced reads on gfx950 (MI355X_MICROARCH.md, HBM section): x2; WRITE_SIZE and narrower accesses are uncalibrated"}, indent=1))
PY
rm -rf $O/*/
