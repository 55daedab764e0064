#!/usr/bin/env python3
"""Where a wave of the force kernel spends its WALL time (diagnostic, instrumented build):
   PEDONI_FORCE_TRACE=1 python tools/force_trace.py [exact|fast]
Every wave adds the shader cycles (s_memtime) between its phase boundaries; shares of the wave
lifetime and cycles per wave are printed for the bench crowd (N = 1e6, rho = 1)."""
import os
import sys
from pathlib import Path

os.environ["PEDONI_FORCE_TRACE"] = "1"
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import numpy as np            # noqa: E402
import bench                  # noqa: E402
from pedoni_amd import abi, host   # noqa: E402

mode = sys.argv[1] if len(sys.argv) > 1 else "exact"
L = 1000.0
obstacles, waypoints = bench.box_geometry(L, L)
field = host.Field.build((L, L), 0.25, obstacles, waypoints)
pos, dest, v0, vel = bench.uniform_crowd(1_000_000, (12.0, L - 12.0), (2.0, L - 2.0), seed=12345)
opt = abi.Options(math_mode=abi.MATH_FAST if mode == "fast" else abi.MATH_EXACT, initial_capacity=1_300_000)
m = abi.HipModel(opt, (L, L), field.distance_map, field.potential_maps, field.unit, obstacles, diagnostics=True)
m.append(pos, dest, v0, vel)
m.tick_n(10)
m.debug_force_trace(reset=True)
steps = 40
m.tick_n(steps)
s = m.debug_force_trace()
names = ["prologue (loads, goal stencil, ranges)", "phase 1 (cutoff test, compaction)", "phase 2 (pair forces)",
         "phase 3 (ordered sums)", "epilogue (wall stencil, integrator, key, counts)"]
waves, life = s[6], s[5]
print(f"{mode}: {waves // steps} waves per launch, {life / waves:.0f} cycles per wave lifetime "
      f"(stamps cost ~10 %)")
for n, c in zip(names, s[:5]):
    print(f"  {n:52s} {c / waves:9.0f} cycles per wave  {100.0 * c / life:5.1f} %")
m.close()
