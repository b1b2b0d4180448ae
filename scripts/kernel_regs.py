#!/usr/bin/env python3
"""Register / spill / LDS figures of the kernels in one object file of llark_amd/csrc/build (no GPU needed).

usage: scripts/kernel_regs.py gemm.o [name-substring ...]
"""
import re
import subprocess
import sys
import tempfile
from pathlib import Path

LLVM = Path("/opt/rocm/lib/llvm/bin")


def notes(obj: Path) -> str:
    with tempfile.TemporaryDirectory() as d:
        fat, dev = Path(d) / "fat.bin", Path(d) / "dev.o"
        subprocess.check_call([LLVM / "llvm-objcopy", "-O", "binary", "--only-section=.hip_fatbin", obj, fat])
        subprocess.check_call([LLVM / "clang-offload-bundler", "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950",
                               f"--input={fat}", f"--output={dev}", "--unbundle"])
        return subprocess.check_output([LLVM / "llvm-readelf", "--notes", dev], text=True)


def main() -> None:
    obj = Path(sys.argv[1])
    if not obj.exists():
        obj = Path(__file__).resolve().parent.parent / "llark_amd" / "csrc" / "build" / sys.argv[1]
    pats = sys.argv[2:]
    for block in notes(obj).split("- .agpr_count")[1:]:
        name = re.search(r"\.name:\s+(\S+)", block).group(1)
        if pats and not any(p in name for p in pats):
            continue
        demangled = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()

        def g(key: str) -> str:
            m = re.search(key + r":\s+(\d+)", block)
            return m.group(1) if m else "?"

        agpr = re.match(r":\s+(\d+)", block).group(1)
        vals = [g("[.]vgpr_count"), agpr, g("[.]vgpr_spill_count"), g("[.]sgpr_count"), g("[.]group_segment_fixed_size")]
        print("vgpr %3s agpr %3s spill %3s sgpr %3s lds %6s  %s" % (*vals, demangled[:150]))


if __name__ == "__main__":
    main()
