"""Parity at BASELINE.json's full sizes (configs[2] C3 and configs[3] C4, N = 1e6): the
oracle is fast enough to check every agent bit for bit, plus size-independent invariants
of the device state (cell order, prefix counts, speed clamp)."""
import sys
from pathlib import Path

import numpy as np
import pytest

from helpers import GOLDEN, bit_equal
from pedoni_amd import scenario as scn

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))


def _check_invariants(gpu, unit=1.4):
    pos, dest, vel, v0 = gpu.download()
    rows, cols = gpu.neighbor_grid_shape()
    idx = gpu.neighbor_grid_indices()
    cx = np.trunc(pos[:, 0] / np.float32(unit)).astype(np.int64)
    cy = np.trunc(pos[:, 1] / np.float32(unit)).astype(np.int64)
    key = cy * cols + cx
    assert (np.diff(key) >= 0).all(), "agents are not in row-major cell order"
    counts = np.bincount(key, minlength=rows * cols)
    assert idx[0] == 0 and np.array_equal(np.diff(idx.astype(np.int64)), counts)
    assert idx[-1] == len(pos)
    return pos, dest, vel, v0


def _run_case(hip, oracle, size, field, obstacles, pos, dest, v0, vel, ticks, **opt):
    ofield = oracle.Field(field.unit, field.distance_map, field.potential_maps)
    cpu = oracle.OracleModel(size, use_distance_map=opt.get("use_distance_map", True))
    gpu = hip.HipModel(hip.Options(initial_capacity=len(pos), **opt), size, field.distance_map,
                       field.potential_maps, field.unit, obstacles)
    cpu.spawn_pedestrians(ofield, pos, dest, v0, vel)
    gpu.append(pos, dest, v0, vel)
    gpu.sort_despawn()
    for t in range(ticks):
        gpu.update_states()
        cpu.update_states(ofield, obstacles)
        gp, gd, gv, g0 = gpu.download()
        wp, wd, wv, w0 = cpu.download()
        assert np.array_equal(gd, wd)
        assert bit_equal(gp, wp).all() and bit_equal(gv, wv).all(), f"tick {t}"
        speed = np.hypot(gv[:, 0].astype(np.float64), gv[:, 1].astype(np.float64))
        ok = ~np.isfinite(speed) | (speed <= 1.3 * g0.astype(np.float64) * (1 + 1e-6))
        assert ok.all(), "speed clamp violated"            # sfm.rs:252
        gpu.sort_despawn()
        cpu.spawn_pedestrians(ofield)
        assert np.array_equal(gpu.neighbor_grid_indices(), cpu.neighbor_grid_indices())
        _check_invariants(gpu)
    gpu.close()


def test_c3_uniform_crowd_1e6(hip, oracle):
    import bench
    from pedoni_amd import host
    L = 1000.0
    obstacles, waypoints = bench.box_geometry(L, L)
    field = host.Field.build((L, L), 0.25, obstacles, waypoints)
    pos, dest, v0, vel = bench.uniform_crowd(1_000_000, (12.0, L - 12.0), (2.0, L - 2.0), seed=12345)
    _run_case(hip, oracle, (L, L), field, obstacles, pos, dest, v0, vel, ticks=3)


@pytest.mark.parametrize("use_distance_map", [True, False])
def test_c4_bottleneck_x5_1e6(hip, oracle, use_distance_map):
    """bottleneck.toml geometry x5, counter-flow halves, both obstacle-force paths."""
    from pedoni_amd import host
    text = (GOLDEN / "scenarios" / "bottleneck_x5.toml").read_text()
    sc = scn.loads(text)
    field = host.Field.from_scenario(host.Scenario(text), 0.25)
    rng = np.random.default_rng(4)
    n = 1_000_000
    pos = np.zeros((0, 2), np.float32)
    while len(pos) < n:                                   # free space only (distance map > 0.6 m)
        p = rng.uniform(2.0, 998.0, (int((n - len(pos)) * 1.3) + 1000, 2)).astype(np.float32)
        iy, ix = (p[:, 1] / 0.25).astype(int), (p[:, 0] / 0.25).astype(int)
        pos = np.concatenate([pos, p[field.distance_map[iy, ix] > 0.6]])[:n]
    dest = (pos[:, 0] < 500.0).astype(np.uint32)          # left half walks right, right half left
    v0 = np.clip(rng.normal(1.34, 0.26, n), 0.5, 2.2).astype(np.float32)
    vel = np.zeros((n, 2), np.float32)
    vel[:, 0] = np.where(dest == 1, 0.5, -0.5) * v0
    _run_case(hip, oracle, sc.field.size, field, sc.obstacle_array(), pos, dest, v0, vel, ticks=2,
              use_distance_map=use_distance_map)


def test_soak_600_ticks_100k_agents(hip, oracle):
    """Long free run: 600 ticks of 100 000 agents with no host read-back in between, then a
    bitwise comparison with the oracle's own 600 ticks.  (Chaos is no obstacle when every
    tick is bit-identical.)  Agents reach their goals and despawn along the way."""
    from helpers import inject_crowd, oracle_field, random_obstacle_scenario
    sc = random_obstacle_scenario(260.0, 400, seed=3)
    field = oracle_field(oracle, sc)
    pos, dest, v0, vel = inject_crowd(field, sc.field.size, 100_000, 4, seed=5)
    cpu = oracle.OracleModel(sc.field.size)
    gpu = hip.HipModel(hip.Options(), sc.field.size, field.distance_map, field.potential_maps,
                       field.unit, sc.obstacle_array())
    cpu.spawn_pedestrians(field, pos, dest, v0, vel)
    gpu.append(pos, dest, v0, vel)
    for chunk in (200, 400):
        gpu.tick_n(chunk)
        for _ in range(chunk):
            cpu.spawn_pedestrians(field)
            cpu.update_states(field)
        gp, gd, gv, g0 = gpu.download()
        wp, wd, wv, w0 = cpu.download()
        assert len(gp) == len(wp) and np.array_equal(gd, wd)
        assert bit_equal(gp, wp).all() and bit_equal(gv, wv).all() and bit_equal(g0, w0).all()
    assert len(gp) < 100_000          # some agents arrived and despawned
    gpu.close()
