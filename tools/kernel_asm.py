#!/usr/bin/env python3
"""Disassembly of one kernel of a built library:  tools/kernel_asm.py <lib.so> <mangled-name-substring> > out.s"""
import re
import subprocess
import sys
import tempfile
from pathlib import Path

LLVM = Path("/opt/rocm/lib/llvm/bin")
lib = Path(sys.argv[1]).resolve()
want = sys.argv[2]
with tempfile.TemporaryDirectory() as tmp:
    fat = Path(tmp) / "fat.bin"
    subprocess.run([str(LLVM / "llvm-objcopy"), "-O", "binary", "--only-section=.hip_fatbin", str(lib), str(fat)], check=True)
    co = Path(tmp) / "co.o"
    subprocess.run([str(LLVM / "clang-offload-bundler"), "--unbundle", "--type=o", f"--input={fat}",
                    "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--output={co}"], check=True)
    syms = subprocess.run([str(LLVM / "llvm-readelf"), "-sW", str(co)], capture_output=True, text=True, check=True).stdout
    names = [ln.split()[-1] for ln in syms.splitlines() if want in ln and " FUNC " in ln]
    if not names:
        sys.exit(f"no kernel matching {want}")
    out = subprocess.run([str(LLVM / "llvm-objdump"), "-d", f"--disassemble-symbols={names[0]}", str(co)],
                         capture_output=True, text=True, check=True).stdout
    print(out)
