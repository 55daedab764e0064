"""The opt-in GPU field builder (pedoni_hip_eikonal, pedoni_field_build_gpu).

NOT a parity path: upstream's heap fast marching (field.rs:118-192) gives numbers that depend
on its pop order, which no parallel solver reproduces.  What is pinned here is what the
solver claims: the fixed point of the first-order upwind update -- checked against an
independent numpy relaxation of the same update run to convergence -- the eikonal
equation's known solutions, closeness to the heap builder's maps, and that a simulation on
GPU-built maps behaves (agents walk to their goal and despawn)."""
import numpy as np
import pytest

from helpers import GOLDEN, random_obstacle_scenario
from pedoni_amd import abi, host
from pedoni_amd import scenario as scn

pytestmark = pytest.mark.gpu
INF = np.float32(1e24)


def _numpy_fixed_point(u0, f):
    """Jacobi relaxation of u = min(u, upwind(neighbours)) to convergence, float32 throughout."""
    u = u0.astype(np.float32).copy()
    f = np.broadcast_to(np.asarray(f, np.float32), u.shape)
    src = u0 == 0
    big = np.float32(3e38)
    for _ in range(20000):
        p = np.pad(u, 1, constant_values=big)
        a = np.minimum(p[1:-1, :-2], p[1:-1, 2:])
        b = np.minimum(p[:-2, 1:-1], p[2:, 1:-1])
        lo, hi = np.minimum(a, b), np.maximum(a, b)
        with np.errstate(invalid="ignore", over="ignore"):
            d = a - b
            quad = (a + b + np.sqrt(np.maximum(np.float32(2) * f * f - d * d, 0).astype(np.float32))) * np.float32(0.5)
            cand = np.where((hi < 1e23) & (hi - lo < f), quad, lo + f)
        cand = np.where(lo < 1e23, cand, big).astype(np.float32)
        new = np.where(src, u, np.minimum(u, cand))
        if np.array_equal(new, u):
            return u
        u = new
    raise AssertionError("numpy relaxation did not converge")


def test_solver_reaches_the_fixed_point_of_the_upwind_update(hip):
    rng = np.random.default_rng(5)
    rows, cols = 150, 210                                   # not multiples of the 16 x 16 tile
    u0 = np.full((rows, cols), INF, np.float32)
    u0[40:44, 30:32] = 0
    u0[120, 190] = 0
    f = np.full((rows, cols), 0.25, np.float32)
    wall = np.zeros((rows, cols), bool)
    wall[70, 20:180] = True
    wall[20:110, 100] = True
    wall[rng.random((rows, cols)) < 0.02] = True
    f[wall] = 0.25e6                                        # field.rs:102: obstacles are slow, not closed
    got, launches = abi.eikonal(u0, f)
    want = _numpy_fixed_point(u0, f)
    assert launches > 8
    assert np.array_equal(got == 0, u0 == 0)
    rel = np.abs(got - want) / np.maximum(want, 1e-6)
    assert rel.max() < 2e-6, rel.max()
    # constant slowness: distance to the zero set, the eikonal solution up to the scheme's error
    got2, _ = abi.eikonal(u0, 0.25)
    yy, xx = np.mgrid[0:rows, 0:cols]
    d_true = np.minimum(np.hypot(np.maximum(0, np.maximum(40 - yy, yy - 43)), np.maximum(0, np.maximum(30 - xx, xx - 31))),
                        np.hypot(yy - 120, xx - 190)) * 0.25
    far = d_true > 2.0
    assert np.all(got2[far] >= d_true[far] * 0.999)         # first-order upwind over-estimates ...
    assert np.max(got2[far] / d_true[far]) < 1.25           # ... by a bounded factor (<= ~sqrt(2) on diagonals, less far out)


def test_gpu_builder_against_the_heap_builder():
    """Same rasterisation bit for bit; maps close to -- not equal to -- the heap pass's (whose
    values depend on its pop order): reported, with loose bars that a wrong solver would miss."""
    sc = scn.load(GOLDEN / "scenarios" / "narrow_gap.toml")
    heap = host.Field.build(sc.field.size, 0.25, sc.obstacle_array(), sc.waypoint_array())
    gpu = host.Field.build(sc.field.size, 0.25, sc.obstacle_array(), sc.waypoint_array(), solver="gpu")
    assert gpu.shape == heap.shape and gpu.n_maps == heap.n_maps and gpu.gpu_launches > 0
    assert np.array_equal(gpu.obstacle_exist, heap.obstacle_exist)
    free = ~heap.obstacle_exist
    for g, h in [(gpu.distance_map, heap.distance_map)] + list(zip(gpu.potential_maps, heap.potential_maps)):
        assert np.array_equal(g == 0, h == 0)
        sel = free & (h > 1.0) & (h < 1e5)
        rel = np.abs(g[sel] - h[sel]) / h[sel]
        assert np.median(rel) < 0.02 and rel.max() < 0.35, (np.median(rel), rel.max())
        assert np.all(g[sel] <= h[sel] * 1.0001)            # the fixed point is the smaller solution


def test_simulation_on_gpu_built_maps_reaches_the_goal(hip):
    """Functional check of the opt-in path end to end: narrow-gap's 50 agents on maps from the
    GPU builder walk through the gap and despawn at the waypoint, as on the heap-built maps."""
    sc = scn.load(GOLDEN / "scenarios" / "narrow_gap.toml")
    counts = {}
    for solver in ("heap", "gpu"):
        field = host.Field.build(sc.field.size, 0.25, sc.obstacle_array(), sc.waypoint_array(), solver=solver)
        m = hip.HipModel(hip.Options(), sc.field.size, field.distance_map, field.potential_maps, field.unit,
                         sc.obstacle_array())
        rng = np.random.default_rng(1)
        pos = np.stack([np.full(50, 3.0), rng.uniform(3.0, 17.0, 50)], 1).astype(np.float32)
        m.append(pos, np.ones(50, np.uint32))
        seen = []
        for _ in range(6):
            m.tick_n(50)
            seen.append(m.get_pedestrian_count())
        counts[solver] = seen
        m.close()
    assert counts["gpu"][0] == 50 and counts["gpu"][-1] < 10, counts
    assert abs(counts["gpu"][-1] - counts["heap"][-1]) <= 10, counts


def test_large_field_start_up(hip):
    """C3-sized map (4000 x 4000 texels): the builder converges inside its launch budget."""
    import time
    import bench
    obstacles, waypoints = bench.box_geometry(1000.0, 1000.0)
    t0 = time.perf_counter()
    f = host.Field.build((1000.0, 1000.0), 0.25, obstacles, waypoints, solver="gpu")
    dt = time.perf_counter() - t0
    assert f.shape == (4000, 4000) and f.gpu_launches > 100
    dm = f.distance_map
    assert 499.0 < float(dm.max()) < 520.0                  # the box centre is ~500 m from the walls
    print(f"GPU field build 4000 x 4000 x 3 maps: {dt:.2f} s, {f.gpu_launches} relaxation launches")
