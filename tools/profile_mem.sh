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
rocprofv3 -L > "$OUT/counters_available.txt" 2>&1 || true
# A counter set the hardware cannot collect aborts rocprofv3, which then may never exit: every pass is
# checked BEFORE it is launched -- each counter must be in `rocprofv3 -L`, and a pass may hold at most
# 2 TA, 2 TD and 4 TCP counters (the per-block limits met on gfx950) -- and is refused otherwise.  The
# timeout below stays as the backstop, not as the mechanism.
check_pass() {
    local name=$1; shift
    local ta=0 td=0 tcp=0 c
    for c in "$@"; do
        if [ -s "$OUT/counters_available.txt" ] && ! grep -qw "$c" "$OUT/counters_available.txt"; then
            echo "pass $name REFUSED: counter $c is not in rocprofv3 -L"; return 1
        fi
        case $c in TA_*) ta=$((ta+1));; TD_*) td=$((td+1));; TCP_*) tcp=$((tcp+1));; esac
    done
    if [ $ta -gt 2 ] || [ $td -gt 2 ] || [ $tcp -gt 4 ]; then
        echo "pass $name REFUSED: $ta TA / $td TD / $tcp TCP counters (limits 2 / 2 / 4 per pass)"; return 1
    fi
}
pass() {
    name=$1; shift
    check_pass "$name" "$@" || return 0
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
