#!/usr/bin/env python3
"""One character per instruction of a kernel's main loop (first to last MFMA), to see where LDS reads sit relative to the MFMAs:
M mfma, R ds_read_b128, t ds_read_b64_tr_b16, r other ds_read, W ds_write, w s_waitcnt, | s_barrier, e v_exp, b branch, D global->LDS
DMA, G other global access, . other VALU, ' ' scalar / other.   usage: scripts/asm_order.py <object in csrc/build> <kernel substring>"""
import re
import subprocess
import sys
import tempfile
from pathlib import Path

LLVM = Path("/opt/rocm/lib/llvm/bin")


def main():
    obj = Path(sys.argv[1])
    if not obj.exists():
        obj = Path(__file__).resolve().parent.parent / "llark_amd" / "csrc" / "build" / sys.argv[1]
    with tempfile.TemporaryDirectory() as d:
        fat, dev = Path(d) / "fat.bin", Path(d) / "dev.o"
        subprocess.check_call([LLVM / "llvm-objcopy", "-O", "binary", "--only-section=.hip_fatbin", obj, fat])
        subprocess.check_call([LLVM / "clang-offload-bundler", "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950",
                               f"--input={fat}", f"--output={dev}", "--unbundle"])
        txt = subprocess.check_output([LLVM / "llvm-objdump", "-d", dev], text=True)
    parts = re.split(r"\n[0-9a-f]+ <([^>]+)>:\n", txt)
    for i in range(1, len(parts), 2):
        name, body = parts[i], parts[i + 1]
        if not all(p in name for p in sys.argv[2:]):
            continue
        lines = [l for l in body.splitlines() if re.match(r"\s+\S+", l)]
        idx = [k for k, l in enumerate(lines) if "v_mfma" in l]
        if not idx:
            continue
        seq = []
        for l in lines[max(0, idx[0] - 40): idx[-1] + 1]:
            op = re.match(r"\s+(\S+)", l).group(1)
            seq.append("M" if op.startswith("v_mfma") else "t" if op.startswith("ds_read_b64_tr") else "R" if op.startswith("ds_read_b128")
                       else "r" if op.startswith("ds_read") else "W" if op.startswith("ds_write") else "w" if op == "s_waitcnt"
                       else "|" if op == "s_barrier" else "e" if op.startswith("v_exp") else "b" if op.startswith("s_cbranch")
                       else "D" if op.startswith("global_load_lds") else "G" if op.startswith("global") else "." if op.startswith("v_") else " ")
        print(subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()[:90], len(lines), "instructions")
        print("".join(seq))


if __name__ == "__main__":
    main()
