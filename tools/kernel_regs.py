#!/usr/bin/env python3
"""Register budget of every kernel in a built library (VGPRs, SGPRs, scratch, LDS, spills), read from
the gfx950 code object's metadata notes:  tools/kernel_regs.py [lib.so] [name-filter]"""
import re
import subprocess
import sys
import tempfile
from pathlib import Path

LLVM = Path("/opt/rocm/lib/llvm/bin")


def main():
    lib = Path(sys.argv[1] if len(sys.argv) > 1 else "pedoni_amd/lib/libpedoni_hip.so").resolve()
    flt = sys.argv[2] if len(sys.argv) > 2 else "force_kernel"
    with tempfile.TemporaryDirectory() as tmp:
        # the fat binary sits in .hip_fatbin; the bundler pulls the gfx950 code object out of it
        fat = Path(tmp) / "fat.bin"
        subprocess.run([str(LLVM / "llvm-objcopy"), "-O", "binary", "--only-section=.hip_fatbin", str(lib), str(fat)], check=True)
        co = Path(tmp) / "co.o"
        subprocess.run([str(LLVM / "clang-offload-bundler"), "--unbundle", "--type=o", f"--input={fat}",
                        "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--output={co}"], check=True)
        notes = subprocess.run([str(LLVM / "llvm-readelf"), "--notes", str(co)], check=True, capture_output=True, text=True).stdout
    rows = []
    for blk in re.split(r"\n\s+- \.agpr_count", notes)[1:]:
        name = re.search(r"\.name:\s+(\S+)", blk)
        if not name or flt not in name.group(1):
            continue
        get = lambda k: (re.search(r"\." + k + r":\s+(\d+)", blk) or [None, "?"])[1]
        dem = subprocess.run(["c++filt", name.group(1)], capture_output=True, text=True).stdout.strip()
        dem = dem.replace("void pedoni::", "").replace("(pedoni::ForceArgs)", "")
        rows.append(f"{dem:60s} vgpr {get('vgpr_count'):>4} sgpr {get('sgpr_count'):>4} scratch {get('private_segment_fixed_size'):>5} "
                    f"lds {get('group_segment_fixed_size'):>6} spills v{get('vgpr_spill_count')}/s{get('sgpr_spill_count')}")
    print("\n".join(sorted(rows)))


if __name__ == "__main__":
    main()
