#!/bin/bash
# The bench lines of every BASELINE config on one GPU (no CPU baseline): bash tools/bench_all.sh TAG
TAG=${1:?tag}; shift; OUT=gpurun_out/bench_$TAG; mkdir -p $OUT
run() { name=$1; shift; python3 bench.py --no-cpu-baseline "$@" > $OUT/$name.json 2> $OUT/$name.err; python3 - $OUT/$name.json $name <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    r = d.get("roofline") or {}
    print(f"{sys.argv[2]:10s} {d['ms_per_step']*1e3:7.1f} us/step  {d['value']/1e9:6.2f} G agent-steps/s  force {r.get('avg_launch_ms', 0)*1e3:6.1f} us ({r.get('timed_launches')} timed)  breakdown(us) { {k: round(v*1e3,1) for k,v in d.get('kernel_ms_per_step',{}).items()} }")
except Exception as e:
    print(sys.argv[2], "failed:", e)
PY
}
run c3 "$@"
run c3_fast --math fast "$@"
run c2 --workload c2 "$@"
run c4 --workload c4 "$@"
run c4seg --workload c4seg "$@"
run c3_noprofile --no-profile "$@"
run c2_noprofile --workload c2 --no-profile "$@"
