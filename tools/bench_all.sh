#!/bin/bash
# every BASELINE configuration at 200 steps on the round's last tree -> gpurun_out/${TAG:-r04_v1}_bench_*.json + _all.txt
OUT=gpurun_out
export TAG=${TAG:-r04_v1}
run() { tag=$1; shift; python bench.py --gpus 1 --steps 200 --warmup 20 --no-cpu-baseline --no-fast-leg "$@" > $OUT/${TAG:-r04_v1}_bench_$tag.json 2> $OUT/${TAG:-r04_v1}_bench_$tag.err; python - $tag <<'PY'
import json,sys
tag=sys.argv[1]
import os
d=json.loads([l for l in open(f"gpurun_out/{os.environ.get('TAG','r04_v1')}_bench_{tag}.json") if l.startswith("{")][-1])
r=d["roofline"]
print(f"{tag:12s} {d['ms_per_step']*1e3:6.1f} us/step  {d['value']/1e9:6.2f} G agent-steps/s  force {r['avg_launch_ms']*1e3:6.1f} us ({r['timed_launches']} timed)  agents {d['config']['agents_total']}  breakdown(us) { {k: round(v*1e3,1) for k,v in d['kernel_ms_per_step'].items()} }")
PY
}
run c3
run c3_fast --math fast
run c2 --workload c2
run c4 --workload c4
run c4seg --workload c4seg
run 8e6 --agents-per-gpu 8000000
