for n in 800000 860000 900000 917000 935000 960000 1000000 1075000 1200000 1376000 1400000; do
  python3 bench.py --agents-per-gpu $n --steps 100 --warmup 10 --no-cpu-baseline --no-fast-leg 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
n=d['config']['agents_total']
print('N=%d live=%d waves/7168=%.3f tick %.1f us force %.1f us  force per 1e6 agents %.1f us' % ($n, n, n/64/7168, d['ms_per_step']*1e3, r['avg_launch_ms']*1e3, r['avg_launch_ms']*1e3/n*1e6))"
done
