#!/bin/bash
# rocprofv3 evidence of ONE bench workload (C4 / C4seg / C2 profiled like C3):
#   bash tools/profile_workload.sh TAG WORKLOAD      e.g.  r03_c4 c4
# -> profiles/TAG_kernel_stats.csv, TAG_pmc_raw.json, TAG_pmc_force.json, TAG_stalls.json, TAG_bench.json
# Every --pmc pass is its own run with --kernel-trace only; the program follows `--` directly.
TAG=${1:?tag}; WL=${2:?workload}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/profw_$TAG
rm -rf "$OUT"; mkdir -p "$OUT/st"
export TMPDIR=/tmp
cd /tmp
B="python3 $ROOT/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-profile --no-fast-leg --workload $WL"
# the stats pass traces the driver's own command (event timing on, so that the run's line can be set beside the trace)
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- python3 $ROOT/bench.py --gpus 1 --steps 20 --warmup 5 \
    --no-cpu-baseline --no-fast-leg --workload $WL > "$OUT/stats.log" 2>&1 || echo "stats pass failed"
grep '^{' "$OUT/stats.log" > "$OUT/stats_bench.json"
echo "stats done"
rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d "$OUT/fetch" -- $B > "$OUT/fetch.log" 2>&1 || echo "fetch pass failed"
rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE -d "$OUT/write" -- $B > "$OUT/write.log" 2>&1 || echo "write pass failed"
echo "traffic done"
pass() {
    name=$1; shift
    rocprofv3 --kernel-trace --output-format csv --pmc "$@" -d "$OUT/st/$name" -- $B > "$OUT/st/$name.log" 2>&1 \
        || echo "pass $name failed (kept going)"
    echo "pass $name done"
}
pass a SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU
pass b SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS
pass d GRBM_GUI_ACTIVE GRBM_COUNT
cd "$ROOT"
python3 bench.py --no-cpu-baseline --workload $WL --steps 100 --warmup 10 > "$OUT/bench.json" 2> "$OUT/bench.err"
python3 tools/pmc_summary.py "$TAG" "$OUT/stats" "$OUT/fetch" "$OUT/write" - "$OUT/bench.json" > "$OUT/pmc_summary.txt"
python3 tools/stall_summary.py "$TAG" "$OUT/st" > "$OUT/stall_summary.txt"
python3 tools/timed_region_stats.py "$TAG" "$OUT/stats" "$OUT/stats_bench.json" > "$OUT/timed_region.txt" || echo "timed-region summary failed"
cp "$OUT/bench.json" profiles/${TAG}_bench.json
mkdir -p gpurun_out/profiles_$TAG && cp profiles/${TAG}_* gpurun_out/profiles_$TAG/
ls gpurun_out/profiles_$TAG
