#!/bin/bash
# lanes per agent of the force kernel (PEDONI_FORCE_GROUP = 1, 2, 4) x queue depth, on one workload:
#   bash tools/group_sweep.sh TAG [bench args, e.g. --workload c2]      -> gpurun_out/TAG_group_sweep.txt
TAG=${1:?tag}; shift
OUT=gpurun_out/${TAG}_group_sweep.txt; : > $OUT
for V in "1 0" "2 6" "2 8" "4 4" "4 6" "4 8"; do
  set -- $V "${@:3}"
  G=$1; S=$2; shift 2
  for r in 1 2; do
    PEDONI_FORCE_GROUP=$G PEDONI_FORCE_GROUP_SLOTS=$S python3 bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-fast-leg $BENCH_ARGS 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('group $G slots $S run $r: %.1f us/step, force %.1f us (%d timed)' % (d['ms_per_step']*1e3, r['avg_launch_ms']*1e3, r['timed_launches']))" | tee -a $OUT
  done
done
