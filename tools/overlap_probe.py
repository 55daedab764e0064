"""Does the sort pass (waits on memory) overlap with the force kernel (waits on VALU issue) when
the GPU has both to run?  Two half-size models (5e5 agents each in a 1000 x 500 m box, rho = 1)
tick concurrently on their own streams from two host threads, against ONE 1e6-agent model:
if the pair's combined agent-steps/s is clearly higher, splitting a tick into row slices that
pipeline place(k+1) under force(k) would pay.   gpurun -- python tools/overlap_probe.py"""
import sys
import threading
import time

import numpy as np

sys.path.insert(0, ".")
import bench  # noqa: E402
from pedoni_amd import abi, host  # noqa: E402


def make(n, width, height, seed):
    obstacles, waypoints = bench.box_geometry(width, height)
    field = host.Field.build((width, height), 0.25, obstacles, waypoints)
    opt = abi.Options(initial_capacity=int(n * 1.3))
    m = abi.HipModel(opt, (width, height), field.distance_map, field.potential_maps, field.unit, obstacles)
    pos, dest, v0, vel = bench.uniform_crowd(n, (20.0, width - 20.0), (2.0, height - 2.0), seed)
    m.append(pos, dest, v0, vel)
    m.tick_n(20)
    m.get_pedestrian_count()
    return m


def run(models, ticks):
    agents = sum(m.get_pedestrian_count() for m in models)
    ths = [threading.Thread(target=lambda m=m: (m.tick_n(ticks), m.get_pedestrian_count())) for m in models]
    t0 = time.perf_counter()
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    dt = time.perf_counter() - t0
    return agents * ticks / dt, dt / ticks * 1e6


one = make(1_000_000, 1000.0, 1000.0, 1)
for _ in range(2):
    v, us = run([one], 200)
print(f"one model, 1e6 agents: {v / 1e9:.2f} G agent-steps/s, {us:.1f} us per tick")
one.close()
halves = [make(500_000, 1000.0, 500.0, 2 + k) for k in range(2)]
for k in range(2):
    v, us = run([halves[k]], 200)
print(f"one half alone, 5e5 agents: {v / 1e9:.2f} G agent-steps/s, {us:.1f} us per tick")
for _ in range(2):
    v, us = run(halves, 200)
print(f"two halves concurrently: {v / 1e9:.2f} G agent-steps/s, {us:.1f} us per tick of both")
