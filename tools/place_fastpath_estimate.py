#!/usr/bin/env python3
"""Would a copy-with-offset fast path pay in the place kernel?  The path applies to a WAVE
(64 consecutive sorted agents) whose agents all keep their cell and whose cells nobody enters or leaves; a
wave with a single slow lane still walks the rank scan.  Counted here on the bench's crowd (C3: uniform,
1 agent / m^2, heading for the right-hand waypoint) at a reduced size, ticked by the CPU oracle (the checker
is only used as a crowd generator here): per tick, the share of agents that change cell, of cells with any
in- or outflow, and of 64-agent waves that are clean throughout.    python tools/place_fastpath_estimate.py"""
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import bench                      # noqa: E402
from oracle import pyoracle       # noqa: E402

L, N, GU = 320.0, 102_400, 1.4
obstacles, waypoints = bench.box_geometry(L, L)
field = pyoracle.field_from_scenario((L, L), 0.25, obstacles, waypoints)
pos, dest, v0, vel = bench.uniform_crowd(N, (12.0, L - 12.0), (2.0, L - 2.0), seed=12345)
m = pyoracle.OracleModel((L, L), threads=8)
m.spawn_pedestrians(field, pos, dest, v0, vel)
cols = int(np.ceil(np.float32(L) / np.float32(GU)))
out = ["# place-kernel fast path: how often would it apply?  C3-type crowd (uniform, 1 agent / m^2), 102 400 agents in a 320 m box,",
       "# ticked by the CPU oracle; tick | agents changing cell | cells with in- or outflow | 64-agent waves entirely clean"]
for tick in range(1, 81):
    p0 = m.download()[0]
    c0 = (p0[:, 1] / np.float32(GU)).astype(np.int64) * cols + (p0[:, 0] / np.float32(GU)).astype(np.int64)
    m.update_states(field)
    p1 = m.download()[0]                                  # same order: not yet re-sorted
    c1 = (p1[:, 1] / np.float32(GU)).astype(np.int64) * cols + (p1[:, 0] / np.float32(GU)).astype(np.int64)
    moved = c0 != c1
    dirty = np.zeros(cols * cols + 1, bool)
    dirty[c0[moved]] = True
    dirty[np.clip(c1[moved], 0, cols * cols)] = True
    lane_slow = moved | dirty[c0]
    n_w = len(c0) // 64
    wave_clean = ~lane_slow[:n_w * 64].reshape(n_w, 64).any(axis=1)
    occupied = np.unique(c0)
    if tick in (1, 5, 20, 40, 60, 80):
        out.append(f"{tick:4d} | {moved.mean():6.1%} | {dirty[occupied].mean():6.1%} | {wave_clean.mean():8.3%}  (agents in a clean cell: {(~lane_slow).mean():.1%})")
    m.spawn_pedestrians(field)
out.append("# a wave of ~30 cells is clean only if every one of them is: with a third of the cells dirty that is ~1e-5;")
out.append("# a per-LANE fast path does not shorten a wave either: it lasts as long as its slowest lane's chain of loads.")
txt = "\n".join(out) + "\n"
(ROOT / "profiles" / "r04_place_fastpath_estimate.txt").write_text(txt)
print(txt)
