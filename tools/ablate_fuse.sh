# (PEDONI_ABLATE exists in the diagnostics build of the library only)
export PEDONI_HIP_LIB=$(pwd)/pedoni_amd/lib/libpedoni_hip_diag.so
for v in "PEDONI_ABLATE=0" "PEDONI_NO_FUSE_KEY=1" "PEDONI_NO_FUSE_KEY=1 PEDONI_ABLATE=7" "PEDONI_ABLATE=7"; do
  for r in 1 2; do
  env $v python3 bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-fast-leg 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('$v: tick %.1f us force %.1f us  breakdown %s' % (d['ms_per_step']*1e3, r['avg_launch_ms']*1e3, {k: round(v*1e3,1) for k,v in d.get('kernel_ms_per_step',{}).items()}))"
  done
done
