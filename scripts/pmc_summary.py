#!/usr/bin/env python3
"""Digest of rocprofv3 --pmc passes (csv output) for a set of kernels: per kernel and grid size the mean of every counter, the mean
duration from the kernel trace of the same pass, and the ratios the MI355X guide derives from them.

    python scripts/pmc_summary.py <dir with one sub-directory per pass> <kernel substring> [<kernel substring> ...]

Counter conventions on gfx950 (checked on gemm256n, profiles/r03_pmc_gemm256n_summary.txt): GRBM_GUI_ACTIVE is summed over the 8
XCDs (per-XCD cycles = value / 8; effective clock = that / duration); SQ_VALU_MFMA_BUSY_CYCLES is summed over the 1024 SIMDs;
SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles summed over waves."""
import collections
import csv
import glob
import sys


def main():
    root, pats = sys.argv[1], sys.argv[2:]
    ctr = collections.defaultdict(lambda: collections.defaultdict(list))
    dur = collections.defaultdict(list)
    for f in sorted(glob.glob(root + "/**/*counter_collection.csv", recursive=True)):
        for row in csv.DictReader(open(f)):
            name = row["Kernel_Name"]
            if not any(p in name for p in pats):
                continue
            key = (name[:72], row["Grid_Size"])
            ctr[key][row["Counter_Name"]].append(float(row["Counter_Value"]))
    for f in sorted(glob.glob(root + "/**/*kernel_trace.csv", recursive=True))[:1]:
        for row in csv.DictReader(open(f)):
            name = row["Kernel_Name"]
            if any(p in name for p in pats):
                if "Grid_Size" in row:
                    grid = row["Grid_Size"]
                else:                                    # the kernel trace gives the grid per dimension
                    grid = str(int(row["Grid_Size_X"]) * int(row["Grid_Size_Y"]) * int(row["Grid_Size_Z"]))
                dur[(name[:72], grid)].append((int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e3)
    for key in sorted(ctr):
        c = {k: sum(v) / len(v) for k, v in ctr[key].items()}
        us = sum(dur[key]) / len(dur[key]) if dur.get(key) else float("nan")
        print(f"{key[0]}  grid {key[1]}  launches {len(next(iter(ctr[key].values())))}  {us:.1f} us (profiled pass)")
        print("   " + "  ".join(f"{k}={v:.4g}" for k, v in sorted(c.items())))
        out = []
        if "GRBM_GUI_ACTIVE" in c:
            xcd_cycles = c["GRBM_GUI_ACTIVE"] / 8.0
            out.append(f"clock {xcd_cycles / us / 1e3:.2f} GHz")
            if "SQ_VALU_MFMA_BUSY_CYCLES" in c:
                out.append(f"MFMA pipe busy {c['SQ_VALU_MFMA_BUSY_CYCLES'] / 1024.0 / xcd_cycles:.1%}")
        if "SQ_WAVE_CYCLES" in c and "SQ_WAIT_ANY" in c:
            out.append(f"wave cycles waiting {c['SQ_WAIT_ANY'] / c['SQ_WAVE_CYCLES']:.1%}")
        if "SQ_INSTS_VALU" in c and "SQ_INSTS_MFMA" in c and c["SQ_INSTS_MFMA"]:
            out.append(f"VALU (incl. MFMA) per MFMA {c['SQ_INSTS_VALU'] / c['SQ_INSTS_MFMA']:.1f}")
        if "SQ_INSTS_LDS" in c and "SQ_INSTS_MFMA" in c and c["SQ_INSTS_MFMA"]:
            out.append(f"LDS instructions per MFMA {c['SQ_INSTS_LDS'] / c['SQ_INSTS_MFMA']:.2f}")
        if "SQ_LDS_BANK_CONFLICT" in c and c.get("SQ_LDS_IDX_ACTIVE"):
            out.append(f"LDS bank-conflict cycles {c['SQ_LDS_BANK_CONFLICT'] / c['SQ_LDS_IDX_ACTIVE']:.1%} of LDS active")
        if out:
            print("   -> " + "; ".join(out))


if __name__ == "__main__":
    main()
