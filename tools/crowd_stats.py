#!/usr/bin/env python3
"""What the force kernel's work looks like on a bench crowd as it evolves (why C4 costs more
than C3): per agent the candidates of its 3 x 3 cells (phase 1 walks them) and the neighbours within the
2 m cutoff (phase 2 evaluates them); per wave the fullest lane (what phase 1 is padded to).
    gpurun -- python tools/crowd_stats.py c3|c4|c2 [ticks ...]"""
import sys
from pathlib import Path

import numpy as np
from scipy.spatial import cKDTree

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import bench                                   # noqa: E402
from pedoni_amd import abi, host               # noqa: E402

wl = sys.argv[1] if len(sys.argv) > 1 else "c3"
marks = [int(x) for x in sys.argv[2:]] or [0, 25, 100, 220]
if wl == "c3":
    L = 1000.0
    obstacles, waypoints = bench.box_geometry(L, L)
    size = (L, L)
    field = host.Field.build(size, 0.25, obstacles, waypoints)
    pos, dest, v0, vel = bench.uniform_crowd(1_000_000, (12.0, L - 12.0), (2.0, L - 2.0), seed=12345)
else:
    obstacles, waypoints, size, crowd, _ = bench.other_workload(wl)
    field = host.Field.build(size, 0.25, obstacles, waypoints)
    pos, dest, v0, vel = crowd(field)
m = abi.HipModel(abi.Options(initial_capacity=int(len(pos) * 1.3)), size, field.distance_map, field.potential_maps,
                 field.unit, obstacles)
m.append(pos, dest, v0, vel)
done = 0
print(f"# {wl}: {len(pos)} agents; tick | agents | candidates/agent mean, p99, max | fullest lane of a wave: mean | "
      f"padding = fullest / mean | neighbours within 2 m: mean, p99")
for t in marks:
    m.tick_n(t - done); done = t
    m.sort_despawn()
    p, _, _, _ = m.download()
    rows, cols = m.neighbor_grid_shape()
    cs = m.neighbor_grid_indices().astype(np.int64)
    cx = np.trunc(p[:, 0] / np.float32(1.4)).astype(np.int64); cy = np.trunc(p[:, 1] / np.float32(1.4)).astype(np.int64)
    cnt = np.zeros(len(p), np.int64)
    x0, x1 = np.maximum(cx - 1, 0), np.minimum(cx + 1, cols - 1)
    for dy in (-1, 0, 1):
        y = cy + dy
        ok = (y >= 0) & (y < rows)
        yy = np.clip(y, 0, rows - 1)
        cnt += np.where(ok, cs[yy * cols + x1 + 1] - cs[yy * cols + x0], 0)
    n64 = len(p) // 64 * 64
    fullest = cnt[:n64].reshape(-1, 64).max(axis=1)
    tree = cKDTree(p.astype(np.float64))
    near = tree.query_ball_point(p.astype(np.float64), 2.0, return_length=True, workers=-1) - 1
    print(f"{t:5d} | {len(p)} | {cnt.mean():.1f}, {np.percentile(cnt, 99):.0f}, {cnt.max()} | {fullest.mean():.1f} | "
          f"{fullest.mean() / cnt.mean():.2f} | {near.mean():.1f}, {np.percentile(near, 99):.0f}")
m.close()
