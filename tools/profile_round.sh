#!/bin/bash
# Collect the rocprofv3 evidence of one stage on a GPU box (run through gpurun from the repo
# root):  bash tools/profile_round.sh TAG
# Kernel times and counters come from SEPARATE runs (--pmc passes never share a run with
# --stats beyond the kernel trace; TCC counters one per pass, see MI355X_MICROARCH.md).
set -e
TAG=${1:?tag}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
B="python3 $ROOT/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-profile --no-fast-leg"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- $B > "$OUT/stats.log" 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d "$OUT/fetch" -- $B > "$OUT/fetch.log" 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE -d "$OUT/write" -- $B > "$OUT/write.log" 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVES SQ_INSTS_VALU SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU -d "$OUT/sq" -- $B > "$OUT/sq.log" 2>&1 || echo "SQ pass failed (kept going)"
cd "$ROOT"
python3 bench.py --no-cpu-baseline > "$OUT/bench.json" 2> "$OUT/bench.err"
python3 tools/pmc_summary.py "$TAG" "$OUT/stats" "$OUT/fetch" "$OUT/write" "$OUT/sq" "$OUT/bench.json"
cp "$OUT/bench.json" profiles/${TAG}_bench.json
mkdir -p gpurun_out/profiles_$TAG && cp profiles/${TAG}_* gpurun_out/profiles_$TAG/
ls gpurun_out/profiles_$TAG
