#!/bin/bash
# Vector-memory pipeline counters of the tick's kernels (address unit TA, L1 TCP, data unit TD): is
# the force kernel's other limit the path its gathers take?  Run through gpurun from the repo root:
#   bash tools/profile_mem.sh TAG [bench args, e.g. --math fast]
# Every --pmc pass is its own run with --kernel-trace only.
TAG=${1:?tag}; shift
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/mem_$TAG
rm -rf "$OUT"; mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
B="python3 $ROOT/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-profile --no-fast-leg $*"
pass() {
    name=$1; shift
    # (a counter set the hardware cannot collect aborts rocprofv3, which then may never exit)
    timeout -k 10 150 rocprofv3 --kernel-trace --output-format csv --pmc "$@" -d "$OUT/$name" -- $B > "$OUT/$name.log" 2>&1 \
        || echo "pass $name failed (kept going)"
    echo "pass $name done"
}
# TA and TD take two counters per pass, TCP four
pass m1 TA_TA_BUSY TA_FLAT_READ_WAVEFRONTS
pass m2 TA_ADDR_STALLED_BY_TC_CYCLES TA_DATA_STALLED_BY_TC_CYCLES
pass m3 TCP_TOTAL_CACHE_ACCESSES TCP_TCC_READ_REQ
pass m4 TCP_PENDING_STALL_CYCLES TCP_TCP_TA_DATA_STALL_CYCLES
pass m5 TCP_GATE_EN1 TD_TD_BUSY
pass m6 GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_VMEM_RD SQ_BUSY_CYCLES
cd "$ROOT"
python3 tools/stall_summary.py "$TAG" "$OUT" mem
mkdir -p gpurun_out/profiles_$TAG && cp profiles/${TAG}_mem.json gpurun_out/profiles_$TAG/
