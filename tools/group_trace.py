#!/usr/bin/env python3
"""Per-wave lifetimes and phase times of the group force kernel on C2 (diagnostics build):
   gpurun -- python tools/group_trace.py [c2|c3small] [group]
One traced launch after 60 warm ticks; prints the distribution of wave lifetimes, start offsets and
where the heaviest waves spend their time."""
import os
import sys
from pathlib import Path

grp = sys.argv[2] if len(sys.argv) > 2 else "2"
os.environ["PEDONI_FORCE_TRACE"] = "1"
os.environ["PEDONI_FORCE_GROUP"] = grp
os.environ["PEDONI_NO_GRAPH"] = "1"
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import numpy as np            # noqa: E402
import bench                  # noqa: E402
from pedoni_amd import abi, host   # noqa: E402

obstacles, waypoints, size, crowd, _ = bench.other_workload("c2")
field = host.Field.build(size, 0.25, obstacles, waypoints)
pos, dest, v0, vel = crowd(field)
warm = abi.HipModel(abi.Options(initial_capacity=130_000), size, field.distance_map, field.potential_maps, field.unit, obstacles)
warm.append(pos, dest, v0, vel); warm.tick_n(60); warm.sort_despawn()
p, d, v, s0 = warm.download(); warm.close()
m = abi.HipModel(abi.Options(initial_capacity=130_000), size, field.distance_map, field.potential_maps, field.unit, obstacles,
                 diagnostics=True)
m.append(p, d, s0, v)
m.sort_despawn()
m.debug_force_trace(reset=True)
m.update_states()
n_waves = (len(p) * int(grp) + 63) // 64
rec = m.debug_force_trace_raw(n_waves + 8)
rec = rec[rec[:, 6] > 0]
# rec[7]: start (low 40 bits) and duration (above) on the 100 MHz clock all XCDs share (s_memtime is per XCD)
rt_start = (rec[:, 7] & np.uint64(0xffffffffff)).astype(np.float64)
rt_life = (rec[:, 7] >> np.uint64(40)).astype(np.float64)
rec = rec.astype(np.float64)
GHZ = float(np.median(rec[:, 5] / np.maximum(rt_life, 1.0)) * 0.1)       # shader cycles per 10 ns, measured
life = rec[:, 5]; start = (rt_start - rt_start.min()) * 10.0 * GHZ
us = lambda c: c / (GHZ * 1e3)
print(f"group {grp}: {len(rec)} waves; wave lifetime us: mean {us(life.mean()):.1f}, p50 {us(np.percentile(life,50)):.1f}, p90 {us(np.percentile(life,90)):.1f}, "
      f"p99 {us(np.percentile(life,99)):.1f}, max {us(life.max()):.1f}; start offsets us: p50 {us(np.percentile(start,50)):.2f}, p90 {us(np.percentile(start,90)):.2f}, "
      f"max {us(start.max()):.2f}; last end {us((start + life).max()):.1f}")
names = ["prologue", "phase 1", "phase 2", "phase 3", "epilogue"]
for label, sel in (("all waves", np.ones(len(life), bool)), ("heaviest 1 %", life >= np.percentile(life, 99)), ("lightest 10 %", life <= np.percentile(life, 10))):
    print(f"  {label:14s}: " + ", ".join(f"{n} {us(rec[sel, k].mean()):.1f}" for k, n in enumerate(names)) + f"  (life {us(life[sel].mean()):.1f} us)")
m.close()
