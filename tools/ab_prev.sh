#!/bin/bash
# interleaved A/B of the working tree's libpedoni_hip.so against the one built from exp/prev_csrc (an earlier commit's sources:
# tools/ab_prev_prepare.sh REV, run where .git is)
TAG=${1:?tag}; shift
ROOT=$(pwd); OUT=$ROOT/gpurun_out/abp_$TAG; mkdir -p "$OUT"
BASE="-O3 -std=c++17 -ffp-contract=off -fno-fast-math -fno-slp-vectorize -mllvm -amdgpu-sched-strategy=max-memory-clause --offload-arch=gfx950 -shared -fPIC -I$ROOT/include -I/opt/rocm/include"
mkdir -p /tmp/abp_0 /tmp/abp_1
cp $ROOT/pedoni_amd/lib/libpedoni_host.so /tmp/abp_0/; cp $ROOT/pedoni_amd/lib/libpedoni_host.so /tmp/abp_1/
cp $ROOT/pedoni_amd/lib/libpedoni_hip.so /tmp/abp_0/libpedoni_hip.so
hipcc $BASE -I$ROOT/exp/prev_csrc -o /tmp/abp_1/libpedoni_hip.so $ROOT/exp/prev_csrc/pedoni_hip.hip -ldl > "$OUT/build_prev.log" 2>&1 || { echo "prev build failed"; exit 1; }
for r in $(seq 1 ${ROUNDS:-4}); do
  for i in 0 1; do
    PEDONI_HIP_LIB=/tmp/abp_$i/libpedoni_hip.so python3 bench.py --steps ${STEPS:-200} --warmup 10 --no-cpu-baseline --no-fast-leg ${BENCH_ARGS:-} > "$OUT/bench_${i}_$r.json" 2>/dev/null
  done
done
python3 - "$OUT" ${ROUNDS:-4} <<'PY'
import json, statistics, sys
out, rounds = sys.argv[1], int(sys.argv[2])
for i, name in ((0, "working tree"), (1, "previous commit")):
    tick, force = [], []
    for r in range(1, rounds + 1):
        try:
            d = json.loads(open(f"{out}/bench_{i}_{r}.json").read().strip().splitlines()[-1])
        except Exception:
            continue
        tick.append(d["ms_per_step"] * 1e3); force.append(d["roofline"]["avg_launch_ms"] * 1e3)
    print(f"{name:16s}: tick median {statistics.median(tick):.1f} us (min {min(tick):.1f}, max {max(tick):.1f}), force median {statistics.median(force):.1f} us, {len(tick)} runs")
PY
