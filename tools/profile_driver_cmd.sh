#!/bin/bash
# The driver's own command under rocprofv3 --kernel-trace --stats (the per-kernel average durations the bench
# line's roofline must agree with), plus the untraced line of the same command:  bash tools/profile_driver_cmd.sh TAG
TAG=${1:?tag}
ROOT=$(pwd); OUT=$ROOT/gpurun_out/driver_$TAG; rm -rf "$OUT"; mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- python3 $ROOT/bench.py --gpus 1 --steps 20 --warmup 5 > "$OUT/bench_traced.json" 2> "$OUT/bench_traced.err"
cd "$ROOT"
python3 bench.py --gpus 1 --steps 20 --warmup 5 > "$OUT/bench.json" 2> "$OUT/bench.err"
cp $(find "$OUT/stats" -name '*_kernel_stats.csv' | head -1) "$OUT/${TAG}_kernel_stats.csv"
python3 tools/timed_region_stats.py "$TAG" "$OUT/stats" "$OUT/bench_traced.json" > /dev/null 2>&1 && cp profiles/${TAG}_timed_region.txt "$OUT/" || true
head -12 "$OUT/${TAG}_kernel_stats.csv"; cat "$OUT/bench.json"
