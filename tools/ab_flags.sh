#!/bin/bash
# A/B of compiler flags (on top of the build's own) for libpedoni_hip.so on a GPU box: builds variants into /tmp and runs
# bench.py (no CPU baseline) with each.   bash tools/ab_flags.sh TAG "flags A" "flags B" ...
TAG=${1:?tag}; shift
ROOT=$(pwd); OUT=$ROOT/gpurun_out/ab_$TAG; mkdir -p "$OUT"
BASE="-O3 -std=c++17 -ffp-contract=off -fno-fast-math -fno-slp-vectorize -mllvm -amdgpu-sched-strategy=max-memory-clause --offload-arch=gfx950 -shared -fPIC -I$ROOT/include -I$ROOT/pedoni_amd/csrc -I/opt/rocm/include"
i=0
for FLAGS in "" "$@"; do
    mkdir -p /tmp/ab_$i; cp $ROOT/pedoni_amd/lib/libpedoni_host.so /tmp/ab_$i/   # resolves its libpedoni_hip.so via $ORIGIN
    LIB=/tmp/ab_$i/libpedoni_hip.so
    hipcc $BASE $FLAGS -o $LIB $ROOT/pedoni_amd/csrc/pedoni_hip.hip -ldl > "$OUT/build_$i.log" 2>&1 || { echo "variant $i ($FLAGS): build failed"; i=$((i+1)); continue; }
    for MODE in exact fast; do
        PEDONI_HIP_LIB=$LIB python3 bench.py --steps 100 --warmup 10 --no-cpu-baseline --math $MODE > "$OUT/bench_${i}_$MODE.json" 2> "$OUT/bench_${i}_$MODE.err"
        python3 - "$OUT/bench_${i}_$MODE.json" "$i" "$FLAGS" "$MODE" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(f"variant {sys.argv[2]} [{sys.argv[3]}] {sys.argv[4]}: {d['ms_per_step']*1e3:.1f} us/step, force {d['roofline']['avg_launch_ms']*1e3:.1f} us, kernels {({k: round(v*1e3,1) for k,v in d.get('kernel_ms_per_step',{}).items()})}")
except Exception as e:
    print("variant", sys.argv[2], sys.argv[4], "failed:", e)
PY
    done
    i=$((i+1))
done
