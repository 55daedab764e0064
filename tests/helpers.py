"""Shared input builders for the parity tests (seeded, build-owned)."""
from __future__ import annotations

from pathlib import Path

import numpy as np

from pedoni_amd import scenario as scn

GOLDEN = Path(__file__).resolve().parent / "golden"


def box_scenario(L: float, wall_w: float = 0.2, margin: float = 10.0) -> scn.Scenario:
    """sparse.toml-style box: four thin border walls, waypoint lines
    `margin` metres inside the left and right edges."""
    sc = scn.Scenario()
    sc.field = scn.FieldConfig((L, L))
    sc.waypoints = [scn.SegmentConfig(((margin, margin), (margin, L - margin))),
                    scn.SegmentConfig(((L - margin, margin), (L - margin, L - margin)))]
    sc.obstacles = [scn.SegmentConfig(((0, 0), (0, L)), wall_w),
                    scn.SegmentConfig(((L, 0), (L, L)), wall_w),
                    scn.SegmentConfig(((0, 0), (L, 0)), wall_w),
                    scn.SegmentConfig(((0, L), (L, L)), wall_w)]
    return sc


def random_obstacle_scenario(L: float = 200.0, n_obs: int = 1000, seed: int = 7) -> scn.Scenario:
    """random.toml-style geometry: L x L, 4 corner waypoints, 4 border
    walls + n_obs short thin random walls.  Build-owned numbers, same statistics."""
    rng = np.random.default_rng(seed)
    sc = scn.Scenario()
    sc.field = scn.FieldConfig((L, L))
    a, b = 0.05 * L, 0.1 * L
    sc.waypoints = [scn.SegmentConfig(((a, b), (b, a))),
                    scn.SegmentConfig(((L - a, b), (L - b, a))),
                    scn.SegmentConfig(((a, L - b), (b, L - a))),
                    scn.SegmentConfig(((L - a, L - b), (L - b, L - a)))]
    sc.obstacles = [scn.SegmentConfig(((0, 0), (0, L)), 0.2),
                    scn.SegmentConfig(((0, L), (L, L)), 0.2),
                    scn.SegmentConfig(((0, 0), (L, 0)), 0.2),
                    scn.SegmentConfig(((L, 0), (L, L)), 0.2)]
    for _ in range(n_obs):
        c = rng.uniform(0.05 * L, 0.95 * L, 2)
        ang = rng.uniform(0, np.pi)
        half = 2.5
        d = np.array([np.cos(ang), np.sin(ang)]) * half
        p0, p1 = c - d, c + d
        sc.obstacles.append(scn.SegmentConfig(((float(p0[0]), float(p0[1])),
                                               (float(p1[0]), float(p1[1]))), 0.2))
    return sc


def oracle_field(oracle, sc: scn.Scenario, unit: float = 0.25):
    return oracle.field_from_scenario(sc.field.size, unit, sc.obstacle_array(), sc.waypoint_array())


def inject_crowd(field, size, n: int, n_dest: int, seed: int = 12345, clearance: float = 0.5,
                 min_potential: float = 1.0):
    """Seeded crowd for state injection: positions uniform over free space
    (distance map > clearance), destination uniform, v0 ~ N(1.34, 0.26) clipped to
    [0.5, 2.2], velocity = 0.5 * v0 * (unit vector of a random heading)."""
    rng = np.random.default_rng(seed)
    pos = np.zeros((0, 2), np.float32)
    dest = np.zeros(0, np.uint32)
    unit = field.unit
    tries = 0
    while len(pos) < n:
        tries += 1
        if tries > 200:            # (a field with no cell that qualifies must fail, not spin)
            raise RuntimeError(f"inject_crowd: only {len(pos)} of {n} positions found in 200 rounds: no free space?")
        m = int((n - len(pos)) * 1.5) + 64
        p = rng.uniform([0.6, 0.6], [size[0] - 0.6, size[1] - 0.6], (m, 2)).astype(np.float32)
        d = rng.integers(0, n_dest, m).astype(np.uint32)
        iy = np.clip((p[:, 1] / unit).astype(int), 0, field.shape[0] - 1)
        ix = np.clip((p[:, 0] / unit).astype(int), 0, field.shape[1] - 1)
        ok = field.distance_map[iy, ix] > clearance
        pot = np.stack([pm[iy, ix] for pm in field.potential_maps], 0)[d, np.arange(m)]
        ok &= (pot > min_potential) & (pot < 1e6)
        pos = np.concatenate([pos, p[ok]])[:n]
        dest = np.concatenate([dest, d[ok]])[:n]
    v0 = np.clip(rng.normal(1.34, 0.26, n), 0.5, 2.2).astype(np.float32)
    ang = rng.uniform(0, 2 * np.pi, n)
    vel = (0.5 * v0[:, None] * np.stack([np.cos(ang), np.sin(ang)], 1)).astype(np.float32)
    return pos, dest, v0, vel


def bits(a: np.ndarray) -> np.ndarray:
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def bit_equal(a: np.ndarray, b: np.ndarray) -> np.ndarray:
    """Element-wise bit equality of f32 arrays, any NaN == any NaN."""
    a = np.ascontiguousarray(a, np.float32)
    b = np.ascontiguousarray(b, np.float32)
    return (bits(a) == bits(b)) | (np.isnan(a) & np.isnan(b))


def rel_close(a, b, rtol=1e-5, floor=1e-3):
    """|a-b| <= rtol * max(|b|, floor); NaN matches NaN."""
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    with np.errstate(invalid="ignore"):
        ok = np.abs(a - b) <= rtol * np.maximum(np.abs(b), floor)
    return ok | (a == b) | (np.isnan(a) & np.isnan(b))
