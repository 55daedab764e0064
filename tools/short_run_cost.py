"""Why the driver's 20-step bench line is ~8 % slower per tick than a 200-step one: one model, 1e6 agents,
the timed region of bench.py replayed under each instrumentation setting (wall clock around tick_n + sync,
median of 9).  Run on the GPU box: python tools/short_run_cost.py"""
import os
import statistics
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from pedoni_amd import abi, host  # noqa: E402
side = 1000.0
obstacles, waypoints = bench.box_geometry(side, side)
field = host.Field.build((side, side), 0.25, obstacles, waypoints)
pos, dest, v0, vel = bench.uniform_crowd(1_000_000, (12.0, side - 12.0), (2.0, side - 2.0), 12345)
m = abi.HipModel(abi.Options(initial_capacity=1_300_000), (side, side), field.distance_map, field.potential_maps, field.unit,
                 obstacles)
m.append(pos, dest, v0, vel)
# bench.py's own sequence on a new model: 5 warm-up ticks, then ONE timed region of 20
m.tick_n(5)
m.synchronize()
for k in range(4):
    m.profile(True, kernels=[abi.K_FORCE], every=3)
    m.kernel_times(reset=True)
    m.synchronize()
    t0 = time.perf_counter()
    m.tick_n(20)
    m.synchronize()
    dt = (time.perf_counter() - t0) / 20 * 1e6
    kt = m.kernel_times(reset=True)["force_integrate"]
    m.profile(False)
    print(f"region {k} after a 5-tick warm-up: {dt:7.2f} us/tick, force kernel {1e3 * kt['total_ms'] / kt['launches']:.2f} us "
          f"x {kt['launches']}", flush=True)


def fresh():
    """the same crowd for every setting: the run ages it (agents arrive and leave)"""
    m.clear()
    m.append(pos, dest, v0, vel)
    m.tick_n(30)
    m.synchronize()



def region(steps, every):
    fresh()
    if every:
        m.profile(True, kernels=[abi.K_FORCE], every=every)
        m.kernel_times(reset=True)
    out = []
    for _ in range(9):
        m.synchronize()
        t0 = time.perf_counter()
        m.tick_n(steps)
        m.synchronize()
        out.append((time.perf_counter() - t0) / steps * 1e6)
    if every:
        m.kernel_times(reset=True)
        m.profile(False)
    return statistics.median(out), min(out)


for steps in (20, 200):
    for every in (0, 1, 2, 3, 5, 9, 0, 1):
        med, lo = region(steps, every)
        print(f"steps {steps:4d}  force kernel event-timed every {every}: {med:7.2f} us/tick median, {lo:7.2f} best, "
              f"{m.get_pedestrian_count()} agents", flush=True)
m.close()
