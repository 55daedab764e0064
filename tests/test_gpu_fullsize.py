"""Parity at BASELINE.json's full sizes (configs[2] C3 and configs[3] C4, N = 1e6): the
oracle is fast enough to check every agent bit for bit, plus size-independent invariants
of the device state (cell order, prefix counts, speed clamp)."""
import functools
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


@functools.lru_cache(maxsize=None)
def _c3_case():
    import bench
    from pedoni_amd import host
    L = 1000.0
    obstacles, waypoints = bench.box_geometry(L, L)
    field = host.Field.build((L, L), 0.25, obstacles, waypoints)
    crowd = bench.uniform_crowd(1_000_000, (12.0, L - 12.0), (2.0, L - 2.0), seed=12345)
    return (L, L), field, obstacles, crowd


@functools.lru_cache(maxsize=None)
def _c4_case():
    """bottleneck.toml geometry x5, counter-flow halves."""
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
    return sc.field.size, field, sc.obstacle_array(), (pos, dest, v0, vel)


def test_c3_uniform_crowd_1e6(hip, oracle):
    size, field, obstacles, (pos, dest, v0, vel) = _c3_case()
    _run_case(hip, oracle, size, field, obstacles, pos, dest, v0, vel, ticks=3)


@pytest.mark.parametrize("use_distance_map", [True, False])
def test_c4_bottleneck_x5_1e6(hip, oracle, use_distance_map):
    """bottleneck.toml geometry x5, counter-flow halves, both obstacle-force paths."""
    size, field, obstacles, (pos, dest, v0, vel) = _c4_case()
    _run_case(hip, oracle, size, field, obstacles, pos, dest, v0, vel, ticks=2,
              use_distance_map=use_distance_map)


def _fast_mode_case(hip, oracle, size, field, obstacles, pos, dest, v0, vel, ticks, **opt):
    """PEDONI_MATH_FAST against the oracle, per step from IDENTICAL state (the GPU's state is
    re-injected from the oracle before every step): every agent within north_star's 1e-5 --
    |dv| <= 1e-5 max(|v'|, |a| dt), |dp| <= 1e-5 |p| -- and the same survivors / cells after
    the next pass."""
    from pedoni_amd import abi
    udm = opt.get("use_distance_map", True)
    ofield = oracle.Field(field.unit, field.distance_map, field.potential_maps)
    cpu = oracle.OracleModel(size, use_distance_map=udm)
    gpu = hip.HipModel(hip.Options(initial_capacity=len(pos), math_mode=abi.MATH_FAST, **opt), size,
                       field.distance_map, field.potential_maps, field.unit, obstacles)
    cpu.spawn_pedestrians(ofield, pos, dest, v0, vel)

    def vec_bad(g, w, floor):
        g, w = g.astype(np.float64), w.astype(np.float64)
        err = np.linalg.norm(g - w, axis=1)
        return ~(err <= 1e-5 * np.maximum(np.linalg.norm(w, axis=1), floor)) & \
            ~(np.isnan(g).any(axis=1) & np.isnan(w).any(axis=1))

    worst = 0.0
    for t in range(ticks):
        wp, wd, wv, w0 = cpu.download()
        gpu.clear()
        gpu.append(wp, wd, w0, wv)                          # identical state, oracle order
        gpu.sort_despawn()
        a_dt = np.linalg.norm(cpu.calc_accelerations(ofield, obstacles).astype(np.float64), axis=1) * 0.1
        cpu.update_states(ofield, obstacles)
        gpu.update_states()
        gp, gd, gv, g0 = gpu.download()
        wp, wd, wv, w0 = cpu.download()
        assert np.array_equal(gd, wd)
        bad = vec_bad(gp, wp, 0.0) | vec_bad(gv, wv, a_dt)
        assert not bad.any(), f"tick {t}: {bad.sum()} of {len(wp)} agents outside 1e-5 in fast mode"
        with np.errstate(invalid="ignore", divide="ignore"):
            rel = np.linalg.norm(gv.astype(np.float64) - wv, axis=1) / np.maximum(np.linalg.norm(wv.astype(np.float64), axis=1), a_dt)
        worst = max(worst, float(np.nanmax(rel)))
        cpu.spawn_pedestrians(ofield)
        gpu.sort_despawn()
        assert gpu.get_pedestrian_count() == cpu.get_pedestrian_count()
        assert np.array_equal(gpu.neighbor_grid_indices(), cpu.neighbor_grid_indices())
    gpu.close()
    return worst


def test_fast_math_c3_1e6_within_1e5(hip, oracle):
    size, field, obstacles, (pos, dest, v0, vel) = _c3_case()
    worst = _fast_mode_case(hip, oracle, size, field, obstacles, pos, dest, v0, vel, ticks=2)
    assert worst <= 1e-5


@pytest.mark.parametrize("use_distance_map", [True, False])
def test_fast_math_c4_1e6_within_1e5(hip, oracle, use_distance_map):
    size, field, obstacles, (pos, dest, v0, vel) = _c4_case()
    worst = _fast_mode_case(hip, oracle, size, field, obstacles, pos, dest, v0, vel, ticks=2,
                            use_distance_map=use_distance_map)
    assert worst <= 1e-5


def test_fast_math_dense_crowd_rho6_within_1e5(hip, oracle):
    """rho = 6 / m^2: ~75 in-range pairs per agent, so 75 approximate terms per sum."""
    from helpers import box_scenario, inject_crowd, oracle_field
    sc = box_scenario(70.0)
    field = oracle_field(oracle, sc)
    n = 6 * 60 * 60
    pos, dest, v0, vel = inject_crowd(field, sc.field.size, n, 2, seed=17, clearance=1.0)
    pos = (5.0 + (pos - 0.6) * (60.0 / 68.8)).astype(np.float32)
    worst = _fast_mode_case(hip, oracle, sc.field.size, field, sc.obstacle_array(), pos, dest, v0, vel, ticks=3)
    assert worst <= 1e-5


@functools.lru_cache(maxsize=None)
def _c5_field():
    """The C5 box (1000 x 8000 m, maps at 0.25 m: 3 x 512 MB), built once for both C5 tests."""
    import bench
    from pedoni_amd import host
    obs, wps = bench.box_geometry(1000.0, 8000.0)
    return obs, wps, host.Field.build((1000.0, 8000.0), 0.25, obs, wps)


def test_c5_8e6_agents_through_the_shard_driver(hip, oracle):
    """C5 on the driver that ships: the 8 bands are 8 `pedoni_shard_*` shards
    ticked by pedoni_shard_local_group_tick_n -- the C-ABI driver `bench.py --gpus N` runs, with
    its exchange done by device copies -- each holding only ITS texel rows of the field maps
    (pedoni_hip_create_rows), cut by AGENT count (pedoni_shard_balanced_bounds, then pushed 3 rows
    off so the re-cut has work), re-cut every 2 ticks.  The crowd is denser towards the middle of
    the field, drifts vertically (agents cross every band edge every tick) and the merged bands
    must equal the ORACLE's single 8e6-agent model bit for bit after 5 ticks."""
    import torch
    import bench
    from pedoni_amd import abi

    G, n_total, ticks, cap, slack = 8, 8_000_000, 5, 8192, 10
    W, H = 1000.0, 8000.0
    obs, wps, field = _c5_field()
    rng = np.random.default_rng(2024)
    pos, dest, v0, vel = bench.uniform_crowd(n_total, (12.0, W - 12.0), (2.0, H - 2.0), 777)
    # density 1 +- 0.35 along y (inverse-CDF of 1 + 0.35 cos), so equal ROWS would not be equal AGENTS
    u = (pos[:, 1].astype(np.float64) - 2.0) / (H - 4.0)
    y = u.copy()
    for _ in range(30):                                    # solve y + 0.35 sin(2 pi y) / (2 pi) = u
        y = u - 0.35 * np.sin(2 * np.pi * y) / (2 * np.pi)
    pos[:, 1] = (2.0 + y * (H - 4.0)).astype(np.float32)
    vel[:, 1] = np.where(rng.random(n_total) < 0.5, 1.1, -1.1).astype(np.float32)

    ofield = oracle.Field(field.unit, field.distance_map, field.potential_maps)
    cpu = oracle.OracleModel((W, H))
    cpu.spawn_pedestrians(ofield, pos, dest, v0, vel)
    wp0 = cpu.download()[0]
    rows = int(np.ceil(np.float32(H) / np.float32(1.4)))
    def row_of(yy):
        with np.errstate(invalid="ignore"):                  # (a few agents go NaN: coincident pairs, as upstream)
            return np.trunc(np.nan_to_num(yy / np.float32(1.4), nan=-1.0)).astype(np.int64)
    row_counts = np.bincount(row_of(wp0[:, 1]), minlength=rows).astype(np.uint32)
    ideal = abi.balanced_bounds(row_counts, G)
    bounds = [ideal[0]] + [b + (3 if k % 2 else -3) for k, b in enumerate(ideal[1:-1])] + [ideal[-1]]
    assert max(np.diff(ideal)) > 1.15 * min(np.diff(ideal)), "the crowd is not lopsided enough to matter"

    stream = torch.cuda.current_stream().cuda_stream
    models, shards = [], []
    band_of = np.searchsorted(np.asarray(bounds[1:-1]), row_of(pos[:, 1]), side="right")
    for r in range(G):
        rows_needed = abi.shard_map_rows(bounds[r], bounds[r + 1], slack, 1.4, field.unit, field.shape[0])
        assert rows_needed[1] - rows_needed[0] < field.shape[0] // 5      # a slice, not the whole 512 MB map
        m = hip.HipModel(hip.Options(initial_capacity=int(n_total / G * 1.3)), (W, H), field.distance_map,
                         field.potential_maps, field.unit, obs, map_rows=rows_needed)
        m.set_stream(stream)
        s = abi.Shard(m, r, G, bounds, cap)
        s.set_rebalance(2, 3, map_slack_rows=slack)
        sel = band_of == r
        m.append(pos[sel], dest[sel], v0[sel], vel[sel])
        s.begin()
        models.append(m); shards.append(s)
    assert sum(s.owned_count() for s in shards) == len(wp0)

    crossed = 0
    for t in range(ticks):
        y0 = cpu.download()[0][:, 1]
        cpu.update_states(ofield, obs)
        y1 = cpu.download()[0][:, 1]
        b_now = np.asarray(bounds[1:-1])
        crossed += int((np.searchsorted(b_now, row_of(y0), side="right") !=
                        np.searchsorted(b_now, row_of(y1), side="right")).sum())
        cpu.spawn_pedestrians(ofield)
    abi.local_group_tick_n(shards, ticks)
    # bands: sort (exchange sort update pack)^T = the oracle's state one sort short: a last
    # exchange + sort (no update) through the public halo entry points lines them up
    words = hip.HipModel.halo_bytes(cap) // 4
    sends = [torch.zeros(words, dtype=torch.int32, device="cuda") for _ in range(G)]
    for m, snd in zip(models, sends):
        m.halo_pack(snd.data_ptr(), cap)
    for r, m in enumerate(models):
        m.halo_unpack(sends[r - 1].data_ptr() if r > 0 else None, sends[r + 1].data_ptr() if r + 1 < G else None, cap)
        m.sort_despawn()
    torch.cuda.synchronize()

    wp, wd, wv, w0 = cpu.download()
    got_parts = [s.download_owned() for s in shards]
    gp, gd, gv, g0 = (np.concatenate([p[k] for p in got_parts]) for k in range(4))
    assert sum(s.owned_count() for s in shards) == len(wp) == len(gp)
    assert n_total - 2000 < len(wp) <= n_total
    assert np.array_equal(gd, wd)
    assert bit_equal(gp, wp).all() and bit_equal(gv, wv).all() and bit_equal(g0, w0).all()
    new_bounds = [shards[r].band()[0] for r in range(G)] + [shards[-1].band()[1]]
    assert new_bounds != bounds, "the bands were never re-cut"
    # the re-cut moved every boundary back towards the balanced cut
    assert sum(abs(a - b) for a, b in zip(new_bounds, ideal)) < sum(abs(a - b) for a, b in zip(bounds, ideal))
    assert crossed > 1500, f"only {crossed} agents changed bands: the exchange was not exercised"
    for s in shards:
        s.close()
    for m in models:
        m.close()


def test_c5_8e6_agents_8_bands(hip, oracle):
    """BASELINE.json configs[4] (C5): 8e6 agents in a 1000 x 8000 m box cut into 8 row bands.
    No 8-GPU node is available to the suite, so the 8 bands are 8 models on ONE device and the
    all-gather is replaced by handing every band all send buffers (same kernels, same lists).
    The merged band state must equal the ORACLE's single 8e6-agent model bit for bit after
    every one of 3 ticks' worth of exchange + sort + update, with a vertical velocity that
    makes thousands of agents cross band edges each tick."""
    import torch
    import bench
    from pedoni_amd import host
    from pedoni_amd.sharded import ShardedModel

    G, n_per, ticks = 8, 1_000_000, 3
    W, H = 1000.0, 1000.0 * G
    obs, wps, field = _c5_field()
    parts = []
    for r in range(G):
        # no gap at the band edges (only the outer walls keep their 2 m): the boundary rows are populated
        p, d, s, v = bench.uniform_crowd(n_per, (12.0, W - 12.0), (r * 1000.0 + (2.0 if r == 0 else 0.0),
                                                                    (r + 1) * 1000.0 - (2.0 if r == G - 1 else 0.0)), 12345 + r)
        v[:, 1] = np.where(np.arange(n_per) % 2 == 0, 1.1, -1.1)     # make agents cross band edges
        parts.append((p, d, s, v))
    pos, dest, v0, vel = (np.concatenate([q[k] for q in parts]) for k in range(4))

    ofield = oracle.Field(field.unit, field.distance_map, field.potential_maps)
    cpu = oracle.OracleModel((W, H))
    cpu.spawn_pedestrians(ofield, pos, dest, v0, vel)

    stream = torch.cuda.current_stream().cuda_stream
    cap = 4096
    words = hip.HipModel.halo_bytes(cap) // 4
    sends = [torch.zeros(words, dtype=torch.int32, device="cuda") for _ in range(G)]
    bands = []
    for r in range(G):
        m = hip.HipModel(hip.Options(initial_capacity=int(n_per * 1.2)), (W, H), field.distance_map,
                         field.potential_maps, field.unit, obs)
        m.set_stream(stream)
        bands.append(ShardedModel(m, r, G, halo_cap=cap, gather=lambda s, rv: None, send=sends[r], recv=sends))
    owner = bands[0].owner_of(pos[:, 1])
    for r, b in enumerate(bands):
        sel = owner == r
        b.load(pos[sel], dest[sel], v0[sel], vel[sel])

    crossed = 0
    for t in range(ticks):
        for b in bands:                                     # one band tick = exchange -> sort -> update
            b.pack()
        for b in bands:
            b.unpack()
            b.model.sort_despawn()
            b.model.update_states()
        y0 = cpu.download()[0][:, 1]
        cpu.update_states(ofield, obs)                      # keeps the order: agent i stays agent i
        y1 = cpu.download()[0][:, 1]
        crossed += int((bands[0].owner_of(y0) != bands[0].owner_of(y1)).sum())
        cpu.spawn_pedestrians(ofield)
    # a final exchange + sort lines the bands up with the oracle's last spawn_pedestrians
    for b in bands:
        b.pack()
    for b in bands:
        b.unpack()
        b.model.sort_despawn()
    torch.cuda.synchronize()
    wp, wd, wv, w0 = cpu.download()
    got_parts = [b.download_owned() for b in bands]
    gp, gd, gv, g0 = (np.concatenate([p[k] for p in got_parts]) for k in range(4))
    assert sum(b.owned_count() for b in bands) == len(wp) == len(gp)
    assert G * n_per - 1000 < len(wp) <= G * n_per          # a few agents went NaN / arrived
    assert np.array_equal(gd, wd)
    assert bit_equal(gp, wp).all() and bit_equal(gv, wv).all() and bit_equal(g0, w0).all()
    assert [b.owned_count() for b in bands] == list(np.bincount(bands[0].owner_of(wp[:, 1]), minlength=G))
    assert crossed > 1000, f"only {crossed} agents changed bands: the exchange was not exercised"
    for b in bands:
        b.model.close()


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
