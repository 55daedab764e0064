"""Build the in-tree native libraries (hipcc cross-compiles gfx950 without a GPU)."""
from __future__ import annotations

import os
import shutil
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
CSRC = ROOT / "pedoni_amd" / "csrc"
LIBDIR = ROOT / "pedoni_amd" / "lib"
INCLUDE = ROOT / "include"

# -ffp-contract=off: the reference (Rust) never fuses a*b+c; parity is bit-level.
# -fno-slp-vectorize: the SLP vectoriser pairs x / y arithmetic into v_pk_*_f32; on gfx950 a packed
# op costs about two plain ones (tools/microbench/valu_issue.hip: v_pk_fma_f32 4.4 cycles against
# 2.6 for v_fma_f32 at 6 waves/SIMD) and the pairing adds register moves: the force kernel is 3 %
# faster without it (tools/ab_flags.sh, round 2).  Same operations, same bits.
# -amdgpu-sched-strategy=max-memory-clause: the scheduler groups the force kernel's gathers (6
# candidate loads per batch, the stencils' texel rows) into clauses; with the memory pipeline a
# co-limiter of that kernel that is worth 2-2.5 % of the tick in both math modes
# (profiles/r02_ab_sched.txt: max-ilp 0, iterative-ilp -14 %, amdgpu-trackers -1 %).  Scheduling
# only: same operations, same bits, same register counts.
HIP_FLAGS = [
    "-O3", "-std=c++17", "-ffp-contract=off", "-fno-fast-math", "-fno-slp-vectorize",
    "-mllvm", "-amdgpu-sched-strategy=max-memory-clause", "--offload-arch=gfx950",
    "-shared", "-fPIC", "-Wall", "-Wextra", "-Wno-unused-value",
]


def _newer(target: Path, sources) -> bool:
    if not target.exists():
        return False
    t = target.stat().st_mtime
    return all(Path(s).stat().st_mtime <= t for s in sources)


def hipcc() -> str:
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not Path(exe).exists():
        raise RuntimeError("hipcc not found: the HIP backend cannot be built")
    return exe


def _hip_deps():
    return [CSRC / "pedoni_hip.hip"] + list(CSRC.glob("*.hpp")) + [INCLUDE / "pedoni_hip.h"]


def _hip_cmd(out: Path, extra=()):
    # (librccl is NOT linked: shard.hpp resolves it with dlopen at first use)
    return [hipcc(), *HIP_FLAGS, *extra, f"-I{INCLUDE}", f"-I{CSRC}", "-I/opt/rocm/include", "-o", str(out),
            str(CSRC / "pedoni_hip.hip"), "-ldl"]


def build_hip(force: bool = False, verbose: bool = False) -> Path:
    """libpedoni_hip.so (the product) and libpedoni_hip_diag.so (the same sources with
    -DPEDONI_DIAGNOSTICS: fault-injection hook, instrumented / ablation force kernels -- loaded by
    tests and tools only).  The two compiles run side by side."""
    LIBDIR.mkdir(parents=True, exist_ok=True)
    out, diag = LIBDIR / "libpedoni_hip.so", LIBDIR / "libpedoni_hip_diag.so"
    jobs = []
    for target, extra in ((out, ()), (diag, ("-DPEDONI_DIAGNOSTICS",))):
        if not force and _newer(target, _hip_deps()):
            continue
        cmd = _hip_cmd(target, extra)
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        jobs.append((cmd, subprocess.Popen(cmd)))
    for cmd, job in jobs:
        if job.wait() != 0:
            raise subprocess.CalledProcessError(job.returncode, cmd)
    return out


def build_host(force: bool = False, verbose: bool = False) -> Path | None:
    """C++ host mirror of pedoni-simulator's Simulator (links against libpedoni_hip)."""
    host_dir = CSRC / "host"
    srcs = sorted(p for p in host_dir.glob("*.cpp") if not p.name.endswith("_main.cpp"))
    if not srcs:
        return None
    out = LIBDIR / "libpedoni_host.so"
    deps = srcs + list(host_dir.glob("*.hpp")) + list(INCLUDE.glob("*.h"))
    if not force and _newer(out, deps):
        return out
    cmd = ["g++", "-O2", "-g", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off",
           "-fno-fast-math", "-Wall", "-Wextra", "-pthread", f"-I{INCLUDE}", f"-I{host_dir}",
           "-o", str(out), *map(str, srcs), f"-L{LIBDIR}", "-lpedoni_hip",
           "-Wl,-rpath,$ORIGIN"]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.run(cmd, check=True)
    return out


def build_cli(force: bool = False, verbose: bool = False) -> Path | None:
    """pedoni-headless: the reference binary's headless mode on the C++ host mirror."""
    src = CSRC / "host" / "headless_main.cpp"
    if not src.exists():
        return None
    bindir = ROOT / "pedoni_amd" / "bin"
    bindir.mkdir(parents=True, exist_ok=True)
    out = bindir / "pedoni-headless"
    deps = [src, CSRC / "host" / "pedoni_host.hpp", LIBDIR / "libpedoni_host.so"]
    if not force and _newer(out, deps):
        return out
    cmd = ["g++", "-O2", "-g", "-std=c++17", "-Wall", "-Wextra", "-pthread", f"-I{INCLUDE}",
           f"-I{CSRC / 'host'}", "-o", str(out), str(src), f"-L{LIBDIR}", "-lpedoni_host",
           "-lpedoni_hip", "-Wl,-rpath,$ORIGIN/../lib"]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.run(cmd, check=True)
    return out


def build_oracle(force: bool = False) -> Path:
    """The CPU oracle is test infrastructure; building the checker is not using it."""
    odir = ROOT / "oracle"
    out = odir / "libpedoni_oracle.so"
    if force and out.exists():
        out.unlink()
    subprocess.run(["make", "-s", "-C", str(odir)], check=True)
    return out


def build_loopback(force: bool = False, verbose: bool = False) -> Path:
    """tests/loopback_rccl: an in-process stand-in for RCCL's send / receive, so that the
    multi-rank driver runs with world > 1 on one GPU.  Test infrastructure like the oracle:
    only the tests load it (PEDONI_RCCL_LIB), never the product by itself."""
    ldir = ROOT / "tests" / "loopback_rccl"
    src, out = ldir / "loopback_rccl.cpp", ldir / "libloopback_rccl.so"
    if not force and _newer(out, [src]):
        return out
    cmd = ["g++", "-O2", "-g", "-std=c++17", "-fPIC", "-shared", "-Wall", "-Wextra", "-Wno-unused-result",
           "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", "-o", str(out), str(src), "-L/opt/rocm/lib",
           "-lamdhip64", "-Wl,-rpath,/opt/rocm/lib", "-pthread"]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.run(cmd, check=True)
    return out


def build_all(force: bool = False, verbose: bool = False) -> None:
    build_hip(force, verbose)
    build_host(force, verbose)
    build_cli(force, verbose)
    build_oracle(force)
    build_loopback(force, verbose)


if __name__ == "__main__":
    build_all(force="--force" in sys.argv, verbose=True)
    print("built:", *sorted(p.name for p in LIBDIR.glob("*.so")))
