#!/usr/bin/env python3
"""Timeline of ONE force-kernel launch from per-wave records (diagnostics build, PEDONI_FORCE_TRACE=1): when
every wave started and ended -- how long the launch runs after its last workgroup was dispatched (the drain),
how many wave slots sit idle in it, and what a heaviest-first order could at best recover.
   gpurun -- python tools/force_timeline.py [c3|c4] [settle-ticks]"""
import os
import sys
from pathlib import Path

work = sys.argv[1] if len(sys.argv) > 1 else "c4"
settle = int(sys.argv[2]) if len(sys.argv) > 2 else 220
os.environ["PEDONI_FORCE_TRACE"] = "1"
os.environ["PEDONI_NO_GRAPH"] = "1"
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import numpy as np            # noqa: E402
import bench                  # noqa: E402
from pedoni_amd import abi, host   # noqa: E402

if work == "c3":
    obstacles, waypoints = bench.box_geometry(1000.0, 1000.0)
    size = (1000.0, 1000.0)
    crowd = lambda field: bench.uniform_crowd(1_000_000, (12.0, 988.0), (2.0, 998.0), seed=12345)
else:
    obstacles, waypoints, size, crowd, _ = bench.other_workload(work)
field = host.Field.build(size, 0.25, obstacles, waypoints)
pos, dest, v0, vel = crowd(field)
warm = abi.HipModel(abi.Options(initial_capacity=1_300_000), size, field.distance_map, field.potential_maps, field.unit, obstacles)
warm.append(pos, dest, v0, vel); warm.tick_n(settle); warm.sort_despawn()
p, d, v, s0 = warm.download(); warm.close()
m = abi.HipModel(abi.Options(initial_capacity=1_300_000), size, field.distance_map, field.potential_maps, field.unit, obstacles,
                 diagnostics=True)
m.append(p, d, s0, v)
m.sort_despawn()
m.update_states(); m.sort_despawn()          # (one warm launch of the traced kernel)
m.debug_force_trace(reset=True)
m.update_states()
n_waves = (len(p) + 63) // 64
rec = m.debug_force_trace_raw(n_waves + 8)
rec = rec[rec[:, 6] > 0]
m.close()
# rec[7]: start (low 40 bits) and duration (above) on the 100 MHz clock all XCDs share; rec[0..5]: shader cycles
rt_start = (rec[:, 7] & np.uint64(0xffffffffff)).astype(np.float64)
rt_life = (rec[:, 7] >> np.uint64(40)).astype(np.float64)
rec = rec.astype(np.float64)
GHZ = float(np.median(rec[:, 5] / np.maximum(rt_life, 1.0)) * 0.1)       # shader cycles per 10 ns
us = lambda c: c / (GHZ * 1e3)
life = rt_life * 10.0 * GHZ                                              # in shader cycles, like the phase columns
start = (rt_start - rt_start.min()) * 10.0 * GHZ
end = start + life
T = end.max()
slots = 1024 * 7
print(f"{work}, crowd {settle} ticks old: {len(rec)} waves on {slots} wave slots (7 per SIMD); launch (first wave start -> last wave end) {us(T):.1f} us at {GHZ} GHz")
print(f"  wave lifetime us: mean {us(life.mean()):.1f}  p50 {us(np.percentile(life, 50)):.1f}  p90 {us(np.percentile(life, 90)):.1f}  p99 {us(np.percentile(life, 99)):.1f}  max {us(life.max()):.1f}")
last_start = start.max()
print(f"  last wave dispatched at {us(last_start):.1f} us; drain after it {us(T - last_start):.1f} us; "
      f"work = sum of lifetimes / slots = {us(life.sum() / slots):.1f} us (the launch at full occupancy throughout)")
# occupancy over time
grid = np.linspace(0, T, 41)
occ = [(np.count_nonzero((start <= t) & (end > t))) / slots for t in grid]
print("  occupancy (fraction of the 7168 slots) at 2.5 % steps of the launch:\n   " + " ".join(f"{o:.2f}" for o in occ))
idle_tail = sum((T - np.maximum(end, last_start)).clip(min=0)) / slots     # slot-time idle after the last dispatch, spread over the slots
print(f"  idle slot-time after the last dispatch = {us(idle_tail):.1f} us of launch (per slot)")
# what the heaviest waves are and where they start
heavy = life >= np.percentile(life, 99)
print(f"  heaviest 1 % of the waves: lifetime {us(life[heavy].mean()):.1f} us, start p10/p50/p90 {us(np.percentile(start[heavy], 10)):.1f} / "
      f"{us(np.percentile(start[heavy], 50)):.1f} / {us(np.percentile(start[heavy], 90)):.1f} us; ending last: wave started at "
      f"{us(start[np.argmax(end)]):.1f} us, lived {us(life[np.argmax(end)]):.1f} us")
# greedy longest-first list schedule of the measured lifetimes on the same slots: the bound a heaviest-first order aims at
import heapq
def makespan(order):
    h = [0.0] * slots
    heapq.heapify(h)
    for w in order:
        t = heapq.heappop(h)
        heapq.heappush(h, t + life[w])
    return max(h)
print(f"  list schedule of the measured lifetimes on {slots} slots: in launch order {us(makespan(np.argsort(start))):.1f} us, "
      f"heaviest first {us(makespan(np.argsort(-life))):.1f} us  (lifetimes taken as fixed: they are not -- co-resident waves share a SIMD)")
names = ["prologue", "phase 1", "phase 2", "phase 3", "epilogue"]
for label, sel in (("all waves", np.ones(len(life), bool)), ("heaviest 1 %", heavy), ("lightest 10 %", life <= np.percentile(life, 10))):
    print(f"  {label:14s}: " + ", ".join(f"{n} {us(rec[sel, k].mean()):.1f}" for k, n in enumerate(names)) + f"  (life {us(life[sel].mean()):.1f} us)")
