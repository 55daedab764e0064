#!/bin/bash
# force-kernel queue depth / residency sweep: PEDONI_FORCE_KERNEL = build:slots, build s94 (94 SGPRs: 7
# workgroups per CU) or default.  bash tools/slots_sweep.sh [bench args]
for S in ${SLOTS_LIST:-s94:6 default:6 s94:5 default:5 s94:4 s94:8 default:8}; do
  for MODE in exact fast; do
    PEDONI_FORCE_KERNEL=$S python3 bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-fast-leg --math $MODE "$@" 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('slots $S $MODE: %.1f us/step, force %.1f us' % (d['ms_per_step']*1e3, r['avg_launch_ms']*1e3))"
  done
done
