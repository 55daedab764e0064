#!/bin/bash
# lanes per agent x crowd size on the C3 box (rho = 1): where the group kernel stops paying.
#   bash tools/group_n_sweep.sh TAG  -> gpurun_out/TAG_group_n_sweep.txt
TAG=${1:?tag}; OUT=gpurun_out/${TAG}_group_n_sweep.txt; : > $OUT
for N in ${N_LIST:-25000 50000 100000 200000 300000 400000 600000}; do
  for G in 1 2 4; do
    PEDONI_FORCE_GROUP=$G python3 bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-fast-leg --agents-per-gpu $N ${BENCH_ARGS:-} 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('N $N group $G: %.1f us/step, force %.1f us' % (d['ms_per_step']*1e3, r['avg_launch_ms']*1e3))" | tee -a $OUT
  done
done
