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
order_used, weight_used = m.tile_order()      # the weights the order was built from (the launch before)
m.update_states()
n_waves = (len(p) + 63) // 64
order, weight = m.tile_order()
if len(order):
    tot = weight.sum()
    bucket = np.where(4.0 * weight * len(order) >= 7.0 * tot, 0, np.where(10.0 * weight * len(order) >= 13.0 * tot, 1, 2))
    print(f"tile order in use: {len(order)} tiles, weights mean {weight.mean():.0f} p50 {np.percentile(weight, 50):.0f} p90 {np.percentile(weight, 90):.0f} "
          f"p99 {np.percentile(weight, 99):.0f} max {weight.max()}; buckets (>= 1.75 mean / >= 1.3 mean / rest): {[int((bucket == k).sum()) for k in range(3)]}")
else:
    print("plain XCD-contiguous order (no tile order)")
rec = m.debug_force_trace_raw(n_waves + 8)
wave_id = np.flatnonzero(rec[:, 6] > 0)
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

# the waves that end last: which tile, when they started, where their time went
last = np.argsort(-end)[:12]
print("  the 12 waves ending last (tile, tile weight, start us, life us, prologue / phase 1 / phase 2 / phase 3 / epilogue us):")
for w in last:
    t = int(wave_id[w]) // 4
    tw = int(weight[t]) if len(order) else -1
    print(f"    tile {t:5d} weight {tw:4d}  start {us(start[w]):6.1f}  life {us(life[w]):5.1f}   " + " / ".join(f"{us(rec[w, k]):.1f}" for k in range(5)))
late = start >= np.percentile(start, 90)
print(f"  waves dispatched in the last 10 %: lifetime mean {us(life[late].mean()):.1f} p90 {us(np.percentile(life[late], 90)):.1f} max {us(life[late].max()):.1f} us; "
      + ", ".join(f"{n} {us(rec[late, k].mean()):.1f}" for k, n in enumerate(names)))

if len(order):
    # does the hardware start workgroups in the order asked for?  start time of XCD 0's i-th workgroup (b = 8 i)
    tile_start = np.full(len(order), np.nan)
    first_wave = wave_id % 4 == 0
    tile_start[(wave_id[first_wave] // 4).astype(int)] = start[first_wave]
    idx = [0, 1, 50, 100, 200, 220, 224, 230, 260, 300, 350, 400, 450, 480]
    print("  XCD 0's i-th workgroup (hardware block 8 i): tile, weight, start us:")
    print("   " + "  ".join(f"i={i}: t{int(order[8 * i])} w{int(weight[order[8 * i]])} {us(tile_start[order[8 * i]]):.1f}" for i in idx if 8 * i < len(order)))

    if len(weight_used) == len(weight):
        d = weight.astype(float) - weight_used.astype(float)
        print(f"  weights the order was built from vs this launch's own: mean {weight_used.mean():.0f} -> {weight.mean():.0f}, |difference| mean {np.abs(d).mean():.1f} "
              f"p99 {np.percentile(np.abs(d), 99):.0f} max {np.abs(d).max():.0f}; correlation {np.corrcoef(weight, weight_used)[0, 1]:.3f}")
        for t in (3758, 3104, 503):
            if t < len(weight): print(f"    tile {t}: used {int(weight_used[t])}, now {int(weight[t])}, started {us(tile_start[t]):.1f} us, position in its XCD's order {int(np.flatnonzero(order == t)[0]) // 8}")
