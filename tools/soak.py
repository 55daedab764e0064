"""One-off long soak (not part of the suite): N agents, T ticks free-running on the GPU, then a
bitwise comparison with the oracle's own T ticks.
    python tools/soak.py [N=200000] [T=3000] [segments | hall | hall_segments]
    (segments: use_distance_map = false; hall: a 700 m hall with 12 walls -- with N >= 4e5 the one-lane force kernel, the
    wall early-out and the heaviest-first workgroup order all take part)"""
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import pedoni_amd as hip                      # noqa: E402
from helpers import bit_equal, inject_crowd, oracle_field, random_obstacle_scenario  # noqa: E402
from oracle import pyoracle as oracle         # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 200_000
T = int(sys.argv[2]) if len(sys.argv) > 2 else 3000
segments = len(sys.argv) > 3 and sys.argv[3] in ("segments", "hall_segments")
hall = len(sys.argv) > 3 and sys.argv[3].startswith("hall")
if hall:
    # an open hall with a few walls: most cells are farther than 21 m from any wall (the wall early-out applies), the
    # crowd is large enough for the one-lane force kernel and the heaviest-first workgroup order (>= 4e5 agents)
    sc = random_obstacle_scenario(700.0, 12, seed=8)
else:
    sc = random_obstacle_scenario(360.0, 30 if segments else 700, seed=8)
field = oracle_field(oracle, sc)
pos, dest, v0, vel = inject_crowd(field, sc.field.size, n, 4, seed=13)
cpu = oracle.OracleModel(sc.field.size, use_distance_map=not segments)
gpu = hip.HipModel(hip.Options(use_distance_map=not segments), sc.field.size, field.distance_map, field.potential_maps, field.unit,
                   sc.obstacle_array())
cpu.spawn_pedestrians(field, pos, dest, v0, vel)
gpu.append(pos, dest, v0, vel)
done = 0
for chunk in (T // 3, T // 3, T - 2 * (T // 3)):
    t0 = time.time()
    gpu.tick_n(chunk)
    gp, gd, gv, g0 = gpu.download()
    t1 = time.time()
    for _ in range(chunk):
        cpu.spawn_pedestrians(field)
        cpu.update_states(field, sc.obstacle_array() if segments else None)
    wp, wd, wv, w0 = cpu.download()
    done += chunk
    same = len(gp) == len(wp) and np.array_equal(gd, wd) and bit_equal(gp, wp).all() and \
        bit_equal(gv, wv).all() and bit_equal(g0, w0).all()
    print(f"tick {done}: {len(wp)} agents left, GPU {t1 - t0:.2f} s, CPU {time.time() - t1:.1f} s, "
          f"bit-identical: {same}", flush=True)
    if not same:
        sys.exit(1)
print("SOAK OK")
