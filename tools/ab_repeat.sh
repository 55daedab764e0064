#!/bin/bash
# Interleaved, repeated A/B of build variants of libpedoni_hip.so on a GPU box: one run of the same
# binary differs from the next by up to +-2.5 % (profiles/r02_ab_sched2.txt), so a change worth 1-2 %
# only shows in the MEDIAN of alternating runs.
#   ROUNDS=5 bash tools/ab_repeat.sh TAG "flags of variant 1" "env NAME=VALUE" ...   (variant 0 = the build's own flags;
#   a variant "env NAME=VALUE ..." is the build's own binary run with that environment; DIAG=1 builds every variant
#   with -DPEDONI_DIAGNOSTICS, for the switches that exist in the diagnostics build only: PEDONI_FORCE_PERSIST, PEDONI_ABLATE)
TAG=${1:?tag}; shift
ROOT=$(pwd); OUT=$ROOT/gpurun_out/abr_$TAG; mkdir -p "$OUT"
[ -n "$DIAG" ] && DIAGFLAG="-DPEDONI_DIAGNOSTICS"
BASE="$DIAGFLAG -O3 -std=c++17 -ffp-contract=off -fno-fast-math -fno-slp-vectorize -mllvm -amdgpu-sched-strategy=max-memory-clause --offload-arch=gfx950 -shared -fPIC -I$ROOT/include -I$ROOT/pedoni_amd/csrc -I/opt/rocm/include"
i=0; N=0
for FLAGS in "" "$@"; do
    mkdir -p /tmp/abr_$i; cp $ROOT/pedoni_amd/lib/libpedoni_host.so /tmp/abr_$i/
    ENVS[$i]=""
    if [[ "$FLAGS" == env\ * ]]; then ENVS[$i]="${FLAGS#env }"; FLAGS=""; fi
    hipcc $BASE $FLAGS -o /tmp/abr_$i/libpedoni_hip.so $ROOT/pedoni_amd/csrc/pedoni_hip.hip -ldl > "$OUT/build_$i.log" 2>&1 || { echo "variant $i ($FLAGS): build failed"; exit 1; }
    echo "variant $i = [$FLAGS] env [${ENVS[$i]}]"
    i=$((i+1)); N=$i
done
for r in $(seq 1 ${ROUNDS:-5}); do
    for i in $(seq 0 $((N-1))); do
        env ${ENVS[$i]} PEDONI_HIP_LIB=/tmp/abr_$i/libpedoni_hip.so python3 bench.py --steps ${STEPS:-100} --warmup 10 --no-cpu-baseline ${BENCH_ARGS:-} > "$OUT/bench_${i}_$r.json" 2> /dev/null
        echo "round $r variant $i done"
    done
done
python3 - "$OUT" $N ${ROUNDS:-5} <<'PY'
import json, statistics, sys
out, n, rounds = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
for i in range(n):
    tick, force, fast = [], [], []
    for r in range(1, rounds + 1):
        try:
            d = json.loads(open(f"{out}/bench_{i}_{r}.json").read().strip().splitlines()[-1])
        except Exception:
            continue
        tick.append(d["ms_per_step"] * 1e3); force.append(d["roofline"]["avg_launch_ms"] * 1e3)
        if d.get("fast_math"): fast.append(d["fast_math"]["ms_per_step"] * 1e3)
    med = lambda v: statistics.median(v) if v else float("nan")
    print(f"variant {i}: tick median {med(tick):.1f} us (min {min(tick):.1f}, max {max(tick):.1f}), force median {med(force):.1f} us, "
          f"fast tick median {med(fast):.1f} us, {len(tick)} runs")
PY
