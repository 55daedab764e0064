# what the fused next-tick key costs inside the force kernel (timing diagnostics; results wrong):
# PEDONI_ABLATE bits 1|2|4 = no stencils, no pairs; 32 = no despawn sampling; 64 = no row counts; 128 = no counts
for v in 7 39 71 135 167; do
  PEDONI_ABLATE=$v python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-fast-leg 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline'] or {}
print('PEDONI_ABLATE=$v: tick %.1f us force %.1f us' % (d['ms_per_step']*1e3, (r.get('avg_launch_ms') or 0)*1e3))"
done
