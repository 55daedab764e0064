#!/bin/bash
# Stall / issue counters of the tick's kernels: where the force kernel's
# wave-cycles go.  Run through gpurun from the repo root:  bash tools/profile_stalls.sh TAG [bench args]
# Every --pmc pass is its own run with --kernel-trace only (never combined with other traces).
# SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles (MI355X_MICROARCH.md).
TAG=${1:?tag}; shift
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/stalls_$TAG
rm -rf "$OUT"; mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
B="python3 $ROOT/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-profile --no-fast-leg $*"
rocprofv3 -L > "$OUT/counters_available.txt" 2>&1 || true
pass() {
    name=$1; shift
    rocprofv3 --kernel-trace --output-format csv --pmc "$@" -d "$OUT/$name" -- $B > "$OUT/$name.log" 2>&1 \
        || echo "pass $name failed (kept going)"
    echo "pass $name done"
}
pass a SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU
pass b SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS
pass c SQ_INSTS_SMEM SQ_INSTS_VMEM_WR SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_INSTS_BRANCH SQ_INSTS_VALU
pass d GRBM_GUI_ACTIVE GRBM_COUNT
pass e SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_LEVEL_WAVES SQ_IFETCH SQ_INSTS_FLAT SQ_INSTS_VALU_TRANS SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_FMA_F64
cd "$ROOT"
python3 tools/stall_summary.py "$TAG" "$OUT"
