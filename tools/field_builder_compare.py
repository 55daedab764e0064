#!/usr/bin/env python3
"""GPU field builder (opt-in, non-parity) against the host heap builder: build times and how far
the maps differ.  python tools/field_builder_compare.py  (needs an MI355X)"""
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import bench                                   # noqa: E402
from pedoni_amd import host, scenario as scn   # noqa: E402

cases = {"narrow_gap (80 x 80)": None, "random.toml (800 x 800)": None, "C3 box (4000 x 4000)": None}
sc = scn.load(ROOT / "tests/golden/scenarios/narrow_gap.toml")
cases["narrow_gap (80 x 80)"] = (sc.field.size, sc.obstacle_array(), sc.waypoint_array())
sc = scn.load(ROOT / "tests/golden/scenarios/random.toml")
cases["random.toml (800 x 800)"] = (sc.field.size, sc.obstacle_array(), sc.waypoint_array())
o, w = bench.box_geometry(1000.0, 1000.0)
cases["C3 box (4000 x 4000)"] = ((1000.0, 1000.0), o, w)
host.Field.build((20.0, 20.0), 0.25, np.zeros((0, 5)), [[1, 1, 1, 2, 1.0]], solver="gpu")   # warm up the device
for name, (size, obs, wps) in cases.items():
    t0 = time.perf_counter(); h = host.Field.build(size, 0.25, obs, wps); th = time.perf_counter() - t0
    t0 = time.perf_counter(); g = host.Field.build(size, 0.25, obs, wps, solver="gpu"); tg = time.perf_counter() - t0
    free = ~h.obstacle_exist
    out = []
    for label, gm, hm in [("distance", g.distance_map, h.distance_map)] + \
            [(f"potential[{k}]", a, b) for k, (a, b) in enumerate(zip(g.potential_maps, h.potential_maps))]:
        sel = free & (hm > 1.0) & (hm < 1e5)
        rel = np.abs(gm[sel] - hm[sel]) / hm[sel]
        out.append(f"{label}: median {np.median(rel):.4f} p99 {np.quantile(rel, 0.99):.4f} max {rel.max():.4f}")
    print(f"{name}: heap {th:.3f} s, gpu {tg:.3f} s ({g.gpu_launches} launches); relative difference of the maps: " + "; ".join(out[:2]))
