"""GPU parity: the HIP path (through the C-ABI) against the CPU oracle.

Bar (north_star): positions / velocities within 1e-5 relative fp32 per step from
identical state.  PEDONI_MATH_EXACT is built to do better -- bit-identical -- so these
tests assert 1e-5 everywhere and bit equality wherever the host libm agrees with the
device's glibc-expf replay (checked first, on this machine).
"""
import ctypes
import ctypes.util

import numpy as np
import pytest

from helpers import (GOLDEN, bit_equal, box_scenario, inject_crowd, oracle_field,
                     random_obstacle_scenario, rel_close)
from pedoni_amd import abi, scenario as scn

pytestmark = pytest.mark.gpu


def _libm_expf(x: np.ndarray) -> np.ndarray:
    libm = ctypes.CDLL(ctypes.util.find_library("m") or "libm.so.6")
    libm.expf.restype = ctypes.c_float
    libm.expf.argtypes = [ctypes.c_float]
    return np.array([libm.expf(float(v)) for v in x], np.float32)


def _make_hip(hip, sc, field, **opt):
    options = hip.Options(**opt)
    return hip.HipModel(options, sc.field.size, field.distance_map, field.potential_maps,
                        field.unit, sc.obstacle_array())


def _assert_state_equal(got, want, what, exact=True):
    gp, gd, gv, g0 = got
    wp, wd, wv, w0 = want
    assert len(gp) == len(wp), f"{what}: survivor count {len(gp)} != {len(wp)}"
    assert np.array_equal(gd, wd), f"{what}: destination order differs"
    assert bit_equal(g0, w0).all(), f"{what}: desired_speed order differs"
    for name, g, w in (("pos", gp, wp), ("vel", gv, wv)):
        ok = rel_close(g, w)
        assert ok.all(), f"{what}: {name} outside 1e-5 for {np.count_nonzero(~ok)} values"
        if exact:
            eq = bit_equal(g, w)
            assert eq.all(), (f"{what}: {name} not bit-identical for {np.count_nonzero(~eq)} of "
                              f"{eq.size} values (max |d| = {np.nanmax(np.abs(g - w)):.3e})")


# ---- device arithmetic primitives -------------------------------------------------------
def test_device_div_sqrt_are_ieee(hip):
    rng = np.random.default_rng(1)
    a = np.concatenate([rng.uniform(-10, 10, 50000), rng.lognormal(0, 20, 50000),
                        [0.0, -0.0, 1e-40, 3e38, np.inf, np.nan]]).astype(np.float32)
    b = np.concatenate([rng.uniform(-10, 10, 50000), rng.lognormal(0, 20, 50000),
                        [0.0, 1.0, 1e-40, 1e-38, 2.0, 1.0]]).astype(np.float32)
    from pedoni_amd import abi
    with np.errstate(all="ignore"):
        assert bit_equal(abi.selftest_math(0, a, b), a / b).all()
        assert bit_equal(abi.selftest_math(1, np.abs(a)), np.sqrt(np.abs(a))).all()


def test_device_sqrt_every_significand(hip):
    """sqrt_core (rsq + one residual step) against IEEE sqrt for EVERY float of [1, 4): both
    exponent parities x all 2^23 significands.  Its operations scale exactly with powers of 4
    inside its domain [2^-96, inf), so this is the whole domain up to scaling (the literal sweep of
    all 1.9e9 floats: tools/microbench/sqrt_variants.hip, tools/exhaustive_sqrt.py); plus a
    strided pass over every binade, the domain's edges and what lies outside (generic path)."""
    from pedoni_amd import abi
    x = np.arange(0x3F800000, 0x40800000, dtype=np.uint32).view(np.float32)
    assert bit_equal(abi.selftest_math(1, x), np.sqrt(x)).all()
    y = np.concatenate([np.arange(0, 0x7F800001, 997, dtype=np.uint32),
                        np.arange(0x0F800000 - 64, 0x0F800000 + 64, dtype=np.uint32),
                        np.arange(0x7F800000 - 64, 0x7F800000 + 2, dtype=np.uint32)]).view(np.float32)
    with np.errstate(all="ignore"):
        want = np.sqrt(y)
    got = abi.selftest_math(1, y)
    assert (bit_equal(got, want) | (np.isnan(got) & np.isnan(want))).all()


def test_device_division_core_is_ieee_in_its_domain(hip):
    """pair_force_hot divides with the compiler's own Newton/residual arithmetic minus its
    rescaling wrap (device_math.hpp div_core): identical to IEEE division wherever the hot
    path uses it -- denominators in [2^-48, 2^20), quotients of magnitude >= 2^-50."""
    from pedoni_amd import abi
    rng = np.random.default_rng(61)
    n = 4_000_000
    d = np.exp2(rng.uniform(-48, 20, n)).astype(np.float32) * rng.choice([-1.0, 1.0], n).astype(np.float32)
    q = np.exp2(rng.uniform(-50, 60, n)) * rng.choice([-1.0, 1.0], n)
    a = (q * d.astype(np.float64)).astype(np.float32)
    keep = np.isfinite(a) & (np.abs(a) >= 2.0 ** -100) & (np.abs(a.astype(np.float64) / d) >= 2.0 ** -50)
    a, d = a[keep], d[keep]
    # plus ordinary magnitudes, where all but a vanishing share of real pairs live
    a2 = rng.normal(0, 1.5, n).astype(np.float32)
    d2 = rng.uniform(1e-3, 8.0, n).astype(np.float32)
    a, d = np.concatenate([a, a2, np.ones(n // 4, np.float32)]), np.concatenate([d, d2, d2[: n // 4]])
    with np.errstate(all="ignore"):
        assert bit_equal(abi.selftest_math(6, a, d), a / d).all()


def test_device_constant_division_is_ieee(hip):
    """x / 0.3f and x / 0.2f by fma(x, zh, x*zl): exhaustively exact on the host for
    2^-100 <= |x| <= 2^100; here the device form incl. its fallback range."""
    from pedoni_amd import abi
    rng = np.random.default_rng(6)
    x = np.concatenate([rng.uniform(-4, 4, 400000), rng.lognormal(0, 30, 100000),
                        -rng.lognormal(0, 30, 100000), rng.integers(0, 2**32, 400000, dtype=np.uint64)
                        .astype(np.uint32).view(np.float32),
                        [0.0, -0.0, 1e-45, -1e-45, 7e-31, 8e-31, 1.2e30, 1.3e30, 3e38, np.inf, -np.inf,
                         np.nan]]).astype(np.float32)
    with np.errstate(all="ignore"):
        assert bit_equal(abi.selftest_math(3, x), x / np.float32(0.3)).all()
        assert bit_equal(abi.selftest_math(4, x), x / np.float32(0.2)).all()


def test_device_f32_as_i32_is_rust_semantics(hip):
    """`as_ivec2` (neighbor_grid.rs:27, sfm.rs:113): truncate toward zero, saturate, NaN -> 0."""
    from pedoni_amd import abi
    rng = np.random.default_rng(12)
    with np.errstate(all="ignore"):
        x = np.concatenate([rng.uniform(-3000, 3000, 200000), rng.uniform(-3e9, 3e9, 50000),
                            rng.integers(0, 2**32, 200000, dtype=np.uint64).astype(np.uint32).view(np.float32)
                            .astype(np.float64),
                            [0.0, -0.0, 0.99999994, -0.99999994, 1.0, -1.0, 2147483520.0, 2147483648.0,
                             -2147483648.0, -2147483904.0, 4e9, -4e9, 1e38, -1e38, np.inf, -np.inf, np.nan]]
                           ).astype(np.float32)
    got = abi.selftest_math(5, x).view(np.int32)
    with np.errstate(all="ignore"):
        want = np.where(np.isnan(x), 0, np.clip(np.trunc(x.astype(np.float64)), -2**31, 2**31 - 1)).astype(np.int64)
    assert np.array_equal(got.astype(np.int64), want)


def test_device_exp_replays_host_libm(hip):
    from pedoni_amd import abi
    rng = np.random.default_rng(2)
    x = np.concatenate([-rng.uniform(0, 30, 100000), -rng.lognormal(0, 3, 20000),
                        [0.0, -0.0, -87.9, -88.1, -103.0, -104.5, -1e12, -np.inf, np.nan]]
                       ).astype(np.float32)
    got = abi.selftest_math(2, x)
    want = _libm_expf(x)
    eq = bit_equal(got, want)
    assert eq.all(), f"device exp differs from host libm on {np.count_nonzero(~eq)} inputs: " \
                     f"{x[~eq][:5]}"


def test_fast_math_primitives_within_budget(hip):
    from pedoni_amd import abi
    rng = np.random.default_rng(3)
    a = rng.uniform(0.01, 10, 100000).astype(np.float32)
    b = rng.uniform(0.01, 10, 100000).astype(np.float32)
    assert np.allclose(abi.selftest_math(0, a, b, abi.MATH_FAST), a / b, rtol=1e-6, atol=0)
    assert np.allclose(abi.selftest_math(1, a, None, abi.MATH_FAST), np.sqrt(a), rtol=1e-6, atol=0)
    x = -rng.uniform(0, 20, 100000).astype(np.float32)
    assert np.allclose(abi.selftest_math(2, x, None, abi.MATH_FAST), np.exp(x.astype(np.float64)),
                       rtol=5e-6, atol=0)


# ---- C1: narrow-gap, 200 lock-step ticks ------------------------------------------------
def test_narrow_gap_200_ticks_lockstep(hip, oracle):
    sc = scn.load(GOLDEN / "scenarios" / "narrow_gap.toml")
    field = oracle_field(oracle, sc)
    rng = oracle.Rng(12345)
    # Simulator::new (lib.rs:37-52): 50 agents on the origin waypoint's line
    w = sc.waypoints[0].line
    u = np.array([rng.f32() for _ in range(50)], np.float32)
    p1, p2 = np.array(w[0], np.float32), np.array(w[1], np.float32)
    pos = (p1[None, :] * (np.float32(1.0) - u)[:, None] + p2[None, :] * u[:, None]).astype(np.float32)
    dest = np.ones(50, np.uint32)
    v0 = np.array([rng.normal_approx(1.34, 0.26) for _ in range(50)], np.float32)

    cpu = oracle.OracleModel(sc.field.size)
    gpu = _make_hip(hip, sc, field)
    cpu.spawn_pedestrians(field, pos, dest, v0, np.zeros((50, 2), np.float32))
    gpu.append(pos, dest, v0, None)
    gpu.sort_despawn()
    _assert_state_equal(gpu.download(), cpu.download(), "after initial spawn")

    counts = []
    for step in range(200):
        cpu.spawn_pedestrians(field)           # Simulator::tick, lib.rs:85 (empty spawn)
        gpu.spawn_pedestrians()
        assert np.array_equal(gpu.neighbor_grid_indices(), cpu.neighbor_grid_indices()), \
            f"step {step}: neighbor_grid_indices differ"
        _assert_state_equal(gpu.download(), cpu.download(), f"step {step} sorted")
        cpu.update_states(field)               # lib.rs:90
        gpu.update_states()
        _assert_state_equal(gpu.download(), cpu.download(), f"step {step} integrated")
        counts.append(gpu.get_pedestrian_count())
    # plumbing check of SURVEY 8(d) C1: agents walk through the gap and despawn at the goal
    assert counts[0] == 50 and counts[-1] < 50
    gpu.close()


def test_tick_n_without_host_sync_keeps_parity_through_despawns(hip, oracle):
    """`tick_n` never reads the live count back, so despawned agents leave stale slots
    behind the live range; those must never re-enter the sort (narrow-gap: all 50 agents
    reach the goal and despawn within 400 ticks)."""
    sc = scn.load(GOLDEN / "scenarios" / "narrow_gap.toml")
    field = oracle_field(oracle, sc)
    rng = np.random.default_rng(8)
    pos = np.stack([np.full(50, 3.0), rng.uniform(3, 17, 50)], 1).astype(np.float32)
    dest = np.ones(50, np.uint32)
    v0 = rng.uniform(1.0, 1.6, 50).astype(np.float32)
    cpu = oracle.OracleModel(sc.field.size)
    gpu = _make_hip(hip, sc, field)
    cpu.spawn_pedestrians(field, pos, dest, v0, np.zeros((50, 2), np.float32))
    gpu.append(pos, dest, v0, None)
    seen = []
    for chunk in (60, 60, 60, 220):
        gpu.tick_n(chunk)
        for _ in range(chunk):
            cpu.spawn_pedestrians(field)
            cpu.update_states(field)
        cpu_state = cpu.download()
        got = gpu.download()
        _assert_state_equal(got, cpu_state, f"after {chunk} more ticks")
        seen.append(len(got[0]))
    assert 0 <= seen[-1] < seen[0] <= 50
    gpu.close()


# ---- C2-style: random obstacles, injected crowd, per step from identical state -----------
@pytest.mark.parametrize("n", [1, 2, 63, 64, 65, 5000, 100_000])
def test_random_crowd_per_step_parity(hip, oracle, n):
    # n = 100 000 is BASELINE.json configs[1] (C2) on the file it names: the geometry of the
    # reference's scenarios/random.toml (200 x 200 m, 4 waypoints, 1004 obstacles; committed as
    # the data fixture tests/golden/scenarios/random.toml) with 100 000 injected agents
    if n == 100_000:
        from helpers import GOLDEN
        from pedoni_amd import scenario as scn
        sc = scn.load(GOLDEN / "scenarios" / "random.toml")
        assert len(sc.obstacles) == 1004 and len(sc.waypoints) == 4 and sc.field.size == (200.0, 200.0)
    else:
        sc = random_obstacle_scenario(200.0, 300)
    field = oracle_field(oracle, sc)
    pos, dest, v0, vel = inject_crowd(field, sc.field.size, n, 4, seed=100 + n)
    cpu = oracle.OracleModel(sc.field.size)
    gpu = _make_hip(hip, sc, field)
    cpu.spawn_pedestrians(field, pos, dest, v0, vel)
    gpu.append(pos, dest, v0, vel)
    gpu.sort_despawn()
    assert np.array_equal(gpu.neighbor_grid_indices(), cpu.neighbor_grid_indices())
    _assert_state_equal(gpu.download(), cpu.download(), "sorted")
    # accelerations alone (sfm.rs:93-241)
    acc_g = gpu.calc_accelerations(cpu.get_pedestrian_count())
    acc_c = cpu.calc_accelerations(field)
    assert rel_close(acc_g, acc_c, floor=1e-2).all()
    assert bit_equal(acc_g, acc_c).all()
    for step in range(3):
        cpu.update_states(field)
        gpu.update_states()
        _assert_state_equal(gpu.download(), cpu.download(), f"step {step} integrated")
        cpu.spawn_pedestrians(field)
        gpu.spawn_pedestrians()
        assert np.array_equal(gpu.neighbor_grid_indices(), cpu.neighbor_grid_indices())
        _assert_state_equal(gpu.download(), cpu.download(), f"step {step} sorted")
    gpu.close()


def test_empty_model(hip, oracle):
    sc = box_scenario(40.0)
    field = oracle_field(oracle, sc)
    gpu = _make_hip(hip, sc, field)
    gpu.spawn_pedestrians()
    gpu.update_states()
    gpu.spawn_pedestrians()
    assert gpu.get_pedestrian_count() == 0
    assert len(gpu.list_pedestrians()) == 0
    assert not gpu.neighbor_grid_indices().any()
    gpu.close()


def test_out_of_grid_nan_and_goal_agents_vanish(hip, oracle):
    """sfm.rs:66-74 + neighbor_grid.rs:27-33: agents outside the grid are never binned,
    NaN positions fail `potential > 0.25`, agents on the goal line despawn."""
    sc = box_scenario(40.0)
    field = oracle_field(oracle, sc)
    pos = np.array([[20, 20], [-3, 5], [5, -0.5], [41, 5], [5, 1e9], [np.nan, 5], [5, np.nan],
                    [30.0, 20.0], [-0.5, 20.0], [10.05, 11.0], [39.9, 39.9]], np.float32)
    dest = np.array([1, 1, 1, 1, 1, 1, 1, 0, 1, 0, 1], np.uint32)
    v0 = np.full(len(pos), 1.3, np.float32)
    vel = np.zeros_like(pos)
    cpu = oracle.OracleModel(sc.field.size)
    gpu = _make_hip(hip, sc, field)
    cpu.spawn_pedestrians(field, pos, dest, v0, vel)
    gpu.append(pos, dest, v0, vel)
    gpu.sort_despawn()
    assert gpu.get_pedestrian_count() == cpu.get_pedestrian_count()
    assert np.array_equal(gpu.neighbor_grid_indices(), cpu.neighbor_grid_indices())
    _assert_state_equal(gpu.download(), cpu.download(), "sorted")
    gpu.close()


def test_coincident_agents_go_nan_then_despawn(hip, oracle):
    """coincident agents give NaN forces; NaN agents survive the
    integrator and are dropped by the next despawn pass."""
    sc = box_scenario(40.0)
    field = oracle_field(oracle, sc)
    pos = np.array([[20, 20], [20, 20], [25, 25]], np.float32)
    dest = np.ones(3, np.uint32)
    v0 = np.full(3, 1.3, np.float32)
    vel = np.zeros_like(pos)
    cpu = oracle.OracleModel(sc.field.size)
    gpu = _make_hip(hip, sc, field)
    cpu.spawn_pedestrians(field, pos, dest, v0, vel)
    gpu.append(pos, dest, v0, vel)
    gpu.sort_despawn()
    cpu.update_states(field)
    gpu.update_states()
    _assert_state_equal(gpu.download(), cpu.download(), "integrated")
    assert np.isnan(gpu.download()[0]).any()
    cpu.spawn_pedestrians(field)
    gpu.spawn_pedestrians()
    assert gpu.get_pedestrian_count() == cpu.get_pedestrian_count() == 1
    gpu.close()


def _pair_fuzz_cases(rng, n):
    """(pos, e, pos_i, vel_i) spanning the hot form's domain, its edges and well beyond."""
    def logu(lo, hi, size):
        return (np.exp2(rng.uniform(lo, hi, size)) * rng.choice([-1.0, 1.0], size)).astype(np.float32)
    ang = rng.uniform(0, 2 * np.pi, n)
    e = np.stack([np.cos(ang), np.sin(ang)], 1).astype(np.float32)
    # a crowd's ordinary pairs
    pos = rng.uniform(1, 200, (n, 2)).astype(np.float32)
    pos_i = (pos + rng.uniform(-2.2, 2.2, (n, 2))).astype(np.float32)
    vel = rng.normal(0, 1.2, (n, 2)).astype(np.float32)
    k = n // 8
    # velocities over 90 binades, exact zeros, overflowing squares, infinities
    vel[:k] = logu(-70, 22, (k, 2))
    vel[k:k + k // 4] = 0.0
    vel[k + k // 4:k + k // 2] = logu(60, 66, (k // 4, 2))
    vel[k + k // 2:k + k // 2 + 8] = np.inf
    # separations over 120 binades around the origin (squares denormal or zero)
    a, b = 2 * k, 3 * k
    pos[a:b] = logu(-80, -20, (k, 2))
    pos_i[a:b] = pos[a:b] + logu(-80, 1, (k, 2))
    # the neighbour's next step lands on the agent: t1 -> 0, t2^2 - (|v| dt)^2 cancels
    a, b = 3 * k, 4 * k
    d = (pos[a:b] - pos_i[a:b]).astype(np.float32)
    vel[a:b] = (d * np.float32(10.0) * (1 + rng.uniform(-1, 1, (k, 1)) * np.exp2(rng.uniform(-24, -2, (k, 1))))).astype(np.float32)
    # axis-aligned pairs (a zero numerator in every division by a length)
    a, b = 4 * k, 4 * k + k // 2
    pos_i[a:b, 0] = pos[a:b, 0]
    vel[a:b, 0] = 0.0
    # b near 26 (x / 0.3 near the exp cut-off) and beyond
    a, b = 5 * k, 6 * k
    vel[a:b] = logu(7, 10, (k, 2))
    # NaN goal direction (an agent on a flat potential), NaN neighbour
    e[6 * k:6 * k + 16] = np.nan
    pos_i[6 * k + 16:6 * k + 32] = np.nan
    return pos, e, pos_i, vel


@pytest.mark.parametrize("seed,one_lane", [(1, False), (2, False), (3, False), (4, False), (5, True), (6, True)])
def test_random_api_call_sequences_stay_in_lockstep(hip, oracle, monkeypatch, seed, one_lane):
    """The C-ABI is a small state machine (appended / sorted / updated; keys and per-cell counts
    fused into the previous update or not; gather or general sort form).  Random legal call
    sequences -- spawns of any size incl. none, repeated passes, single updates, un-synced
    batches, full-state appends, clears -- must keep the model bit-identical to the oracle
    driven by the equivalent calls.  `one_lane`: the one-lane-per-agent force kernel (what crowds of 4e5 agents
    and more run), i.e. with the heaviest-first workgroup order rebuilt by every pass, in an open hall where the
    wall early-out applies, and with un-synced batches long enough for the captured runs of 4 / 8 / 16 ticks."""
    if one_lane:
        monkeypatch.setenv("PEDONI_FORCE_GROUP", "1")
        sc = random_obstacle_scenario(110.0, 4, seed=seed)
    else:
        sc = random_obstacle_scenario(70.0, 60, seed=seed)
    field = oracle_field(oracle, sc)
    rng = np.random.default_rng(100 + seed)
    cpu = oracle.OracleModel(sc.field.size, seed=77)
    gpu = _make_hip(hip, sc, field, seed=77, initial_capacity=1)
    pos, dest, v0, vel = inject_crowd(field, sc.field.size, 4000, 4, seed=seed)
    cpu.spawn_pedestrians(field, pos, dest, v0, vel)
    gpu.append(pos, dest, v0, vel)
    gpu.sort_despawn()
    is_sorted = True

    def newcomers(k):
        p, d, s, v = inject_crowd(field, sc.field.size, k, 4, seed=int(rng.integers(1 << 30)))
        return p, d, s, v

    def check(what):
        assert gpu.get_pedestrian_count() == cpu.get_pedestrian_count(), what
        _assert_state_equal(gpu.download(), cpu.download(), what)

    log = []
    for step in range(60):
        op = rng.choice(["spawn", "spawn0", "update", "tick_n", "append", "clear", "acc", "check"],
                        p=[0.2, 0.15, 0.2, 0.15, 0.1, 0.05, 0.05, 0.1])
        log.append(op)
        if op == "spawn":                      # spawn_pedestrians with arrivals (speeds drawn inside)
            p, d, _, _ = newcomers(int(rng.integers(1, 300)))
            cpu.spawn_pedestrians(field, p, d)
            gpu.spawn_pedestrians(p, d)
            is_sorted = True
        elif op == "spawn0":                   # the bare sort/despawn pass, possibly twice in a row
            cpu.spawn_pedestrians(field)
            gpu.spawn_pedestrians()
            is_sorted = True
        elif op == "update":
            if not is_sorted:
                with pytest.raises(Exception, match="sort/despawn pass"):
                    gpu.update_states()
                continue
            cpu.update_states(field)
            gpu.update_states()
            is_sorted = False
        elif op == "tick_n":
            k = int(rng.integers(1, 6)) if not one_lane else int(rng.choice([1, 2, 4, 5, 9, 17, 22]))
            gpu.tick_n(k)
            for _ in range(k):
                cpu.spawn_pedestrians(field)
                cpu.update_states(field)
            is_sorted = False
        elif op == "append":                   # full-state injection, then a pass
            p, d, s, v = newcomers(int(rng.integers(1, 500)))
            gpu.append(p, d, s, v)
            gpu.sort_despawn()
            cpu.spawn_pedestrians(field, p, d, s, v)
            is_sorted = True
        elif op == "clear":
            gpu.clear()
            assert gpu.get_pedestrian_count() == 0
            cpu = oracle.OracleModel(sc.field.size, seed=77)
            gpu.set_speed_rng(77)              # a fresh oracle restarts its desired-speed stream
            p, d, s, v = newcomers(1500)
            gpu.append(p, d, s, v)
            gpu.sort_despawn()
            cpu.spawn_pedestrians(field, p, d, s, v)
            is_sorted = True
        elif op == "acc":
            if is_sorted and cpu.get_pedestrian_count():
                want = cpu.calc_accelerations(field)
                got = gpu.calc_accelerations(len(want))
                assert bit_equal(got, want).all(), f"step {step} {log[-8:]}"
        else:
            check(f"step {step} after {log[-8:]}")
        if is_sorted:
            assert np.array_equal(gpu.neighbor_grid_indices(), cpu.neighbor_grid_indices()), f"step {step} {log[-8:]}"
    check(f"end, {log}")
    gpu.close()


def test_field_stencil_fuzz_patch_and_per_tap_forms(hip, oracle):
    """sobel_filter + bilinear (util.rs:44-75): the device evaluates the 3 x 3 taps from one
    4 x 4 texel patch and falls back to the literal per-tap form when rounding of p +- 1 moves
    a tap off the patch.  Points just below / at / above integers, around the map's borders,
    far outside, huge, negative, NaN and inf must all give the reference's bits."""
    from pedoni_amd import abi
    rng = np.random.default_rng(1618)
    rows, cols = 57, 83
    grid = rng.normal(0, 3, (rows, cols)).astype(np.float32)
    grid[rng.random((rows, cols)) < 0.05] = 1e12          # obstacle-like texels
    grid[5, 7], grid[6, 7] = np.float32(3.4028235e38), np.float32(1e24)
    n = 300_000
    px = rng.uniform(-6, cols + 6, n).astype(np.float32)
    py = rng.uniform(-6, rows + 6, n).astype(np.float32)
    k = n // 6
    # within a few ulps of an integer: p + 1 and p - 1 round across it
    ints = rng.integers(-3, cols + 3, k).astype(np.float32)
    px[:k] = np.nextafter(ints, ints + rng.choice([-1, 1], k).astype(np.float32)).astype(np.float32)
    py[:k] = (rng.integers(-3, rows + 3, k) + rng.choice([0.0, 0.5, 0.99999994, 1e-7], k)).astype(np.float32)
    px[k:2 * k] = (rng.integers(-3, cols + 3, k) + rng.choice([0.0, 0.99999994, 0.9999999, 1e-8], k)).astype(np.float32)
    # magnitudes where p +- 1 is no longer exact, and beyond i32
    big = (np.exp2(rng.uniform(20, 40, k)) * rng.choice([-1.0, 1.0], k)).astype(np.float32)
    px[2 * k:3 * k] = big
    py[3 * k:3 * k + k // 2] = (np.exp2(rng.uniform(20, 40, k // 2)) * rng.choice([-1.0, 1.0], k // 2)).astype(np.float32)
    special = np.array([np.nan, np.inf, -np.inf, -0.0, 0.0, -1.0, -0.99999994, 2147483648.0, -2147483904.0,
                        cols - 1.0, cols - 2.0, cols - 1.0000001, 16777216.0, 16777215.0, 8388607.5], np.float32)
    px[4 * k:4 * k + len(special)] = special
    py[4 * k + len(special):4 * k + 2 * len(special)] = special
    with np.errstate(all="ignore"):
        want_g, want_c = oracle.sample_many(grid, px, py)
        got_g, got_c = abi.selftest_field(grid, px, py)
    bad = np.flatnonzero(~(bit_equal(got_g, want_g).all(axis=1) & bit_equal(got_c, want_c)))
    assert len(bad) == 0, (f"{len(bad)} points differ, e.g. ({px[bad[0]]!r}, {py[bad[0]]!r}): "
                           f"{got_g[bad[0]]} {got_c[bad[0]]} != {want_g[bad[0]]} {want_c[bad[0]]}")


def test_pair_force_fuzz_hot_and_generic_forms_match_the_oracle(hip, oracle):
    """sfm.rs:130-153 pair by pair, 2 M random pairs including every edge of the hot form's
    validity range (device_math.hpp pair_force_hot): the device function both force kernels
    call must equal the oracle bit for bit, whichever form evaluates the pair."""
    from pedoni_amd import abi
    rng = np.random.default_rng(2718)
    pos, e, pos_i, vel = _pair_fuzz_cases(rng, 2_000_000)
    acc0 = rng.normal(0, 3, pos.shape).astype(np.float32)
    with np.errstate(all="ignore"):
        want = oracle.pair_forces(pos, e, pos_i, vel, acc0)
        got = abi.selftest_pair(pos, e, pos_i, vel, acc0)
    eq = bit_equal(got, want).all(axis=1)
    bad = np.flatnonzero(~eq)
    assert len(bad) == 0, (f"{len(bad)} pairs differ, e.g. #{bad[0]}: pos {pos[bad[0]]} pos_i {pos_i[bad[0]]} "
                           f"vel {vel[bad[0]]} e {e[bad[0]]}: {got[bad[0]]} != {want[bad[0]]}")
    # the fuzz did reach both forms: plenty of finite forces, and NaN / untouched ones
    changed = ~bit_equal(want, acc0).all(axis=1)
    assert changed.sum() > 500_000 and np.isnan(want).any() and (~changed).sum() > 10_000
    # fast mode on the ordinary pairs: within its budget
    sl = slice(7 * (len(pos) // 8), None)
    fast = abi.selftest_pair(pos[sl], e[sl], pos_i[sl], vel[sl], acc0[sl], math_mode=abi.MATH_FAST)
    f_ref = want[sl].astype(np.float64) - acc0[sl]
    err = np.linalg.norm(fast.astype(np.float64) - want[sl], axis=1)
    ok = err <= 2e-5 * np.maximum(np.linalg.norm(f_ref, axis=1), 1e-3) + 1e-6 * np.linalg.norm(acc0[sl], axis=1)
    assert ok.all(), f"{(~ok).sum()} fast-mode pairs outside budget"


def test_pair_force_outside_the_hot_range_takes_the_generic_path(hip, oracle):
    """The exact pair force has a hot form valid for sqrt arguments in [2^-96, inf) and b < 26
    (device_math.hpp pair_force_hot) and a generic form for everything else.  Neighbours
    with zero, denormal-square, huge and overflowing velocities, and separations whose
    square is denormal or underflows, must still match the oracle bit for bit."""
    sc = box_scenario(40.0)
    field = oracle_field(oracle, sc)
    rng = np.random.default_rng(77)
    n = 600
    pos = rng.uniform(18.0, 22.0, (n, 2)).astype(np.float32)          # ~37 agents / m^2
    tiny = np.array([[1e-3, 1e-3], [1e-3 + 1e-12, 1e-3], [1e-3, 1e-3 + 3e-20], [1.0001e-3, 1e-3],
                     [2e-23, 1e-23], [3e-23, 1e-23]], np.float32)      # d^2 normal, tiny, denormal, 0
    pos = np.concatenate([pos, tiny])
    vel = rng.normal(0, 1.0, pos.shape).astype(np.float32)
    special = np.array([[0, 0], [1e-25, 0], [1e-20, 1e-20], [250.0, 0], [600.0, -600.0], [1e10, 1e10],
                        [1e19, 0], [3e19, 3e19], [np.inf, 0], [-1e30, 1e5]], np.float32)
    for k, v in enumerate(special):
        vel[k::40][: 15] = v
    vel[n:] = [[0, 0], [1e-25, 0], [0.3, 0.1], [600.0, 0], [0, 0], [1e-30, 0]]
    dest = np.ones(len(pos), np.uint32)
    v0 = np.full(len(pos), 1.3, np.float32)
    cpu = oracle.OracleModel(sc.field.size)
    gpu = _make_hip(hip, sc, field)
    cpu.spawn_pedestrians(field, pos, dest, v0, vel)
    gpu.append(pos, dest, v0, vel)
    gpu.sort_despawn()
    with np.errstate(all="ignore"):
        want = cpu.calc_accelerations(field)
        got = gpu.calc_accelerations(cpu.get_pedestrian_count())
        assert bit_equal(got, want).all(), f"{np.count_nonzero(~bit_equal(got, want))} components differ"
        cpu.update_states(field)
        gpu.update_states()
        _assert_state_equal(gpu.download(), cpu.download(), "integrated")
    gpu.close()


def test_many_agents_in_one_cell_keep_insertion_order(hip, oracle):
    """Stable in-cell order (sfm.rs:67-68) with far more cell-mates than a wavefront."""
    sc = box_scenario(40.0)
    field = oracle_field(oracle, sc)
    rng = np.random.default_rng(5)
    n = 700
    pos = (np.array([21.0, 21.0]) + rng.uniform(0.01, 1.3, (n, 2))).astype(np.float32)
    dest = rng.integers(0, 2, n).astype(np.uint32)
    v0 = rng.uniform(1, 1.6, n).astype(np.float32)
    vel = rng.uniform(-0.5, 0.5, (n, 2)).astype(np.float32)
    cpu = oracle.OracleModel(sc.field.size)
    gpu = _make_hip(hip, sc, field)
    cpu.spawn_pedestrians(field, pos, dest, v0, vel)
    gpu.append(pos, dest, v0, vel)
    gpu.sort_despawn()
    _assert_state_equal(gpu.download(), cpu.download(), "sorted")
    cpu.update_states(field)
    gpu.update_states()
    _assert_state_equal(gpu.download(), cpu.download(), "integrated")
    gpu.close()


# ---- the two device forms of the sort/despawn pass -----------------------------------------
def test_gather_and_general_sort_forms_agree(hip, oracle, monkeypatch):
    """Steady-state gather form vs the atomic general form: same cell index, same order."""
    sc = random_obstacle_scenario(150.0, 100)
    field = oracle_field(oracle, sc)
    pos, dest, v0, vel = inject_crowd(field, sc.field.size, 40_000, 4, seed=44)
    states = {}
    for general in ("0", "1"):
        monkeypatch.setenv("PEDONI_SORT_GENERAL", general)
        gpu = _make_hip(hip, sc, field)
        gpu.append(pos, dest, v0, vel)
        gpu.tick_n(6)
        gpu.sort_despawn()
        states[general] = (gpu.download(), gpu.neighbor_grid_indices())
        gpu.close()
    (a, ia), (b, ib) = states["0"], states["1"]
    assert np.array_equal(ia, ib)
    for x, y in zip(a, b):
        assert bit_equal(x.astype(np.float32), y.astype(np.float32)).all()
    cpu = oracle.OracleModel(sc.field.size)
    cpu.spawn_pedestrians(field, pos, dest, v0, vel)
    for _ in range(6):
        cpu.update_states(field)
        cpu.spawn_pedestrians(field)
    _assert_state_equal(a, cpu.download(), "after 6 ticks")


def test_far_movers_and_spawns_switch_to_general_form(hip, oracle):
    """Agents that jump several cells in one tick (huge injected speed) and agents spawned
    mid-run must still land in the reference's order: the device flag selects the general
    form for exactly those ticks."""
    sc = box_scenario(80.0)
    field = oracle_field(oracle, sc)
    pos, dest, v0, vel = inject_crowd(field, sc.field.size, 6000, 2, seed=45)
    v0[:50] = 60.0                      # 1.3 * 60 m/s * 0.1 s = 7.8 m per tick: > 5 cells
    vel[:50] = [55.0, 10.0]
    cpu = oracle.OracleModel(sc.field.size, seed=5)
    gpu = _make_hip(hip, sc, field, seed=5)
    cpu.spawn_pedestrians(field, pos, dest, v0, vel)
    gpu.append(pos, dest, v0, vel)
    gpu.sort_despawn()
    rng = np.random.default_rng(46)
    for step in range(8):
        cpu.update_states(field)
        gpu.update_states()
        if step in (2, 3, 6):           # Simulator::tick with periodic spawners (lib.rs:70-85)
            new = rng.uniform(20, 60, (37, 2)).astype(np.float32)
            nd = rng.integers(0, 2, 37).astype(np.uint32)
            cpu.spawn_pedestrians(field, new, nd)
            gpu.spawn_pedestrians(new, nd)
        else:
            cpu.spawn_pedestrians(field)
            gpu.spawn_pedestrians()
        assert np.array_equal(gpu.neighbor_grid_indices(), cpu.neighbor_grid_indices()), step
        _assert_state_equal(gpu.download(), cpu.download(), f"step {step} sorted")
    gpu.close()


def test_capacity_growth_mid_run_keeps_order_and_state(hip, oracle):
    """The SoA arrays double when a spawn does not fit (all device pointers change); the
    sorted order, the keys and the stale-slot bookkeeping must survive it."""
    sc = box_scenario(120.0)
    field = oracle_field(oracle, sc)
    rng = np.random.default_rng(31)
    cpu = oracle.OracleModel(sc.field.size, seed=9)
    gpu = _make_hip(hip, sc, field, seed=9, initial_capacity=1)      # 1024-agent minimum
    total = 0
    for step in range(14):
        n_new = int(rng.integers(200, 900))
        new = rng.uniform(15, 105, (n_new, 2)).astype(np.float32)
        nd = rng.integers(0, 2, n_new).astype(np.uint32)
        cpu.spawn_pedestrians(field, new, nd)
        gpu.spawn_pedestrians(new, nd)
        total += n_new
        assert np.array_equal(gpu.neighbor_grid_indices(), cpu.neighbor_grid_indices()), step
        if step % 3 == 2:
            gpu.tick_n(2)
            for _ in range(2):
                cpu.update_states(field)
                cpu.spawn_pedestrians(field)
            gpu.sort_despawn()
        else:
            cpu.update_states(field)
            gpu.update_states()
        _assert_state_equal(gpu.download(), cpu.download(), f"step {step}")
    assert total > 4096                                               # several doublings happened
    gpu.close()


def test_dense_crowd_six_per_square_metre(hip, oracle):
    """rho = 6 / m^2 (a crush): ~100 candidates and ~75 in-range pairs per agent, many
    batches per wave in the force kernel, ~12 agents per cell in the sort."""
    sc = box_scenario(70.0)
    field = oracle_field(oracle, sc)
    n = 6 * 60 * 60
    pos, dest, v0, vel = inject_crowd(field, sc.field.size, n, 2, seed=17, clearance=1.0)
    pos = (5.0 + (pos - 0.6) * (60.0 / 68.8)).astype(np.float32)      # squeeze into 60 m x 60 m
    cpu = oracle.OracleModel(sc.field.size)
    gpu = _make_hip(hip, sc, field)
    cpu.spawn_pedestrians(field, pos, dest, v0, vel)
    gpu.append(pos, dest, v0, vel)
    gpu.sort_despawn()
    for step in range(4):
        cpu.update_states(field)
        gpu.update_states()
        _assert_state_equal(gpu.download(), cpu.download(), f"step {step} integrated")
        cpu.spawn_pedestrians(field)
        gpu.spawn_pedestrians()
        assert np.array_equal(gpu.neighbor_grid_indices(), cpu.neighbor_grid_indices())
    gpu.close()


@pytest.mark.parametrize("work_size", [64, 128, 1024])
def test_gpu_work_size_option(hip, oracle, work_size):
    """SimulatorOptions.gpu_work_size (lib.rs:121,132): workgroup size of the one-lane-per-
    agent force kernel (brute-force path); results do not depend on it."""
    sc = random_obstacle_scenario(50.0, 20)
    field = oracle_field(oracle, sc)
    pos, dest, v0, vel = inject_crowd(field, sc.field.size, 900, 4, seed=work_size)
    cpu = oracle.OracleModel(sc.field.size, use_neighbor_grid=False)
    gpu = _make_hip(hip, sc, field, use_neighbor_grid=False, gpu_work_size=work_size)
    cpu.spawn_pedestrians(field, pos, dest, v0, vel)
    gpu.append(pos, dest, v0, vel)
    gpu.sort_despawn()
    cpu.update_states(field)
    gpu.update_states()
    _assert_state_equal(gpu.download(), cpu.download(), "integrated")
    gpu.close()


def test_invalid_options_are_rejected(hip, oracle):
    from pedoni_amd import abi
    sc = box_scenario(30.0)
    field = oracle_field(oracle, sc)
    with pytest.raises(abi.PedoniError, match="multiple of 64"):
        _make_hip(hip, sc, field, gpu_work_size=100)
    with pytest.raises(abi.PedoniError, match="neighbor_grid_unit"):
        _make_hip(hip, sc, field, neighbor_grid_unit=0.0)
    gpu = _make_hip(hip, sc, field)
    with pytest.raises(abi.PedoniError, match="sort/despawn pass"):
        gpu.update_states()                       # Simulator::tick order: spawn first (lib.rs:85,90)
    gpu.close()


# ---- option paths ------------------------------------------------------------------------
def test_no_neighbor_grid_bruteforce_path(hip, oracle):
    """use_neighbor_grid = false: filter-only despawn (sfm.rs:78-88), O(N^2) pairs (:157-185)."""
    sc = random_obstacle_scenario(60.0, 30)
    field = oracle_field(oracle, sc)
    pos, dest, v0, vel = inject_crowd(field, sc.field.size, 1500, 4, seed=9)
    pos[7] = [-4.0, 3.0]  # outside the field: potential is 1e12-ish -> survives upstream too
    cpu = oracle.OracleModel(sc.field.size, use_neighbor_grid=False)
    gpu = _make_hip(hip, sc, field, use_neighbor_grid=False)
    cpu.spawn_pedestrians(field, pos, dest, v0, vel)
    gpu.append(pos, dest, v0, vel)
    gpu.sort_despawn()
    _assert_state_equal(gpu.download(), cpu.download(), "filtered")
    for step in range(2):
        cpu.update_states(field)
        gpu.update_states()
        _assert_state_equal(gpu.download(), cpu.download(), f"step {step}")
        cpu.spawn_pedestrians(field)
        gpu.spawn_pedestrians()
    gpu.close()


def test_explicit_wall_segment_path(hip, oracle):
    """use_distance_map = false: sfm.rs:193-236 + util.rs:92-103 over every obstacle."""
    sc = random_obstacle_scenario(80.0, 60)
    sc.obstacles.append(scn.SegmentConfig(((40, 40), (40, 40)), 1.0))   # degenerate segment
    sc.obstacles.append(scn.SegmentConfig(((30, 30), (50, 35)), 6.0))   # wide: agents inside
    field = oracle_field(oracle, sc)
    pos, dest, v0, vel = inject_crowd(field, sc.field.size, 4000, 4, seed=11, clearance=0.0)
    cpu = oracle.OracleModel(sc.field.size, use_distance_map=False)
    gpu = _make_hip(hip, sc, field, use_distance_map=False)
    cpu.spawn_pedestrians(field, pos, dest, v0, vel)
    gpu.append(pos, dest, v0, vel)
    gpu.sort_despawn()
    for step in range(2):
        cpu.update_states(field, sc.obstacle_array())
        gpu.update_states()
        _assert_state_equal(gpu.download(), cpu.download(), f"step {step}")
        cpu.spawn_pedestrians(field)
        gpu.spawn_pedestrians()
    gpu.close()


def test_trait_spawn_draws_speeds_like_the_host_mirror(hip, oracle):
    """spawn_pedestrians (trait form): velocity 0 and desired speed drawn inside the model
    (sfm.rs:49-56) from the build-owned seeded generator."""
    sc = box_scenario(40.0)
    field = oracle_field(oracle, sc)
    pos = np.array([[12, 12], [14, 20], [20, 30]], np.float32)
    dest = np.array([1, 1, 1], np.uint32)
    cpu = oracle.OracleModel(sc.field.size, seed=777)
    gpu = _make_hip(hip, sc, field, seed=777)
    cpu.spawn_pedestrians(field, pos, dest)
    gpu.spawn_pedestrians(pos, dest)
    _assert_state_equal(gpu.download(), cpu.download(), "spawned")
    peds = gpu.list_pedestrians()
    assert len(peds) == 3 and set(peds["destination"]) == {1}
    gpu.close()


def test_fast_math_mode_within_1e5(hip, oracle):
    """PEDONI_MATH_FAST: hardware rcp/rsq/sqrt/exp, but exact decisions (goal direction exact;
    field-of-view halving and cancelling pairs recomputed exactly).  Bar, for EVERY agent,
    per step from identical state: |dv| <= 1e-5 * max(|v'|, |a| dt) -- relative to the larger
    of the two terms of v' = v + a dt, since an agent braking to a halt has |v'| << |a| dt --
    and |dp| <= 1e-5 * |p|."""
    from pedoni_amd import abi
    for seed, n, L, n_obs in ((21, 50_000, 200.0, 300), (22, 120_000, 160.0, 100)):
        sc = random_obstacle_scenario(L, n_obs)
        field = oracle_field(oracle, sc)
        pos, dest, v0, vel = inject_crowd(field, sc.field.size, n, 4, seed=seed)
        cpu = oracle.OracleModel(sc.field.size)
        gpu = _make_hip(hip, sc, field, math_mode=abi.MATH_FAST)
        cpu.spawn_pedestrians(field, pos, dest, v0, vel)
        gpu.append(pos, dest, v0, vel)
        gpu.sort_despawn()
        a_dt = np.linalg.norm(cpu.calc_accelerations(field).astype(np.float64), axis=1) * 0.1
        cpu.update_states(field)
        gpu.update_states()
        gp, gd, gv, g0 = gpu.download()
        wp, wd, wv, w0 = cpu.download()
        assert np.array_equal(gd, wd)

        def vec_bad(g, w, floor):
            g, w = g.astype(np.float64), w.astype(np.float64)
            err = np.linalg.norm(g - w, axis=1)
            return ~(err <= 1e-5 * np.maximum(np.linalg.norm(w, axis=1), floor)) & \
                ~(np.isnan(g).any(axis=1) & np.isnan(w).any(axis=1))

        bad = vec_bad(gp, wp, 0.0) | vec_bad(gv, wv, a_dt)
        assert not bad.any(), f"{bad.sum()} of {n} agents outside 1e-5 in fast mode"
        # and the next pass bins / despawns the same agents
        cpu.spawn_pedestrians(field)
        gpu.spawn_pedestrians()
        assert gpu.get_pedestrian_count() == cpu.get_pedestrian_count()
        gpu.close()


@pytest.mark.parametrize("field_unit,grid_unit", [(0.25, 1.4), (0.5, 2.0), (0.3, 1.4), (0.2, 1.0)])
def test_field_and_grid_units_pow2_and_not(hip, oracle, field_unit, grid_unit):
    """`pos / unit` (field.rs:236, neighbor_grid.rs:27): a power-of-two field unit takes the
    multiply form on the device, any other unit the IEEE division; same bits either way."""
    sc = random_obstacle_scenario(60.0, 40)
    field = oracle_field(oracle, sc, unit=field_unit)
    pos, dest, v0, vel = inject_crowd(field, sc.field.size, 8000, 4, seed=91)
    cpu = oracle.OracleModel(sc.field.size, neighbor_grid_unit=grid_unit)
    gpu = _make_hip(hip, sc, field, neighbor_grid_unit=grid_unit, field_grid_unit=field_unit)
    cpu.spawn_pedestrians(field, pos, dest, v0, vel)
    gpu.append(pos, dest, v0, vel)
    gpu.sort_despawn()
    for step in range(4):
        cpu.update_states(field)
        gpu.update_states()
        cpu.spawn_pedestrians(field)
        gpu.spawn_pedestrians()
        assert np.array_equal(gpu.neighbor_grid_indices(), cpu.neighbor_grid_indices()), step
        _assert_state_equal(gpu.download(), cpu.download(), f"step {step}")
    gpu.close()


def test_queue_and_simple_force_kernels_agree(hip, oracle, monkeypatch):
    """The wave-queue force kernel and the one-lane-per-agent kernel are independent
    implementations of sfm.rs:93-241; both must reproduce the oracle bit for bit."""
    sc = random_obstacle_scenario(120.0, 100)
    field = oracle_field(oracle, sc)
    pos, dest, v0, vel = inject_crowd(field, sc.field.size, 30_000, 4, seed=33)
    cpu = oracle.OracleModel(sc.field.size)
    cpu.spawn_pedestrians(field, pos, dest, v0, vel)
    want = cpu.calc_accelerations(field)
    for simple in ("0", "1"):
        monkeypatch.setenv("PEDONI_FORCE_SIMPLE", simple)
        gpu = _make_hip(hip, sc, field)
        gpu.append(pos, dest, v0, vel)
        gpu.sort_despawn()
        got = gpu.calc_accelerations(len(want))
        assert bit_equal(got, want).all(), f"PEDONI_FORCE_SIMPLE={simple}"
        gpu.close()


def test_create_destroy_does_not_leak_device_memory(hip, oracle):
    """pedoni_hip_destroy must return everything pedoni_hip_create, capacity growth, sharding,
    profiling and the spawners allocated."""
    import torch
    sc = random_obstacle_scenario(80.0, 40)
    field = oracle_field(oracle, sc)
    pos, dest, v0, vel = inject_crowd(field, sc.field.size, 20_000, 4, seed=3)

    def cycle(k):
        gpu = _make_hip(hip, sc, field, initial_capacity=1)
        if k % 2:
            gpu.set_band(10, 40, 2048)
        gpu.append(pos, dest, v0, vel)        # grows the arrays several times
        gpu.sort_despawn()
        if not k % 2:
            gpu.profile(True)
            gpu.tick_n(3)
            gpu.kernel_times(reset=True)
        gpu.close()

    cycle(0); cycle(1)                        # warm the allocator's own pools
    torch.cuda.synchronize()
    free0, _ = torch.cuda.mem_get_info()
    for k in range(12):
        cycle(k)
    torch.cuda.synchronize()
    free1, _ = torch.cuda.mem_get_info()
    assert free0 - free1 < 8 << 20, f"{(free0 - free1) / 2**20:.1f} MiB of device memory not returned"


@pytest.mark.parametrize("env", [{"PEDONI_FORCE_GROUP": "1"}, {"PEDONI_FORCE_GROUP": "2"}, {"PEDONI_FORCE_GROUP": "4"},
                                 {"PEDONI_FORCE_GROUP": "2", "PEDONI_FORCE_GROUP_SLOTS": "4"},
                                 {"PEDONI_FORCE_GROUP": "4", "PEDONI_FORCE_GROUP_SLOTS": "8"},
                                 {"PEDONI_FORCE_KERNEL": "s94:6"}, {"PEDONI_FORCE_KERNEL": "s94:4"},
                                 {"PEDONI_FORCE_KERNEL": "default:8"}, {"PEDONI_FORCE_KERNEL": "default:5"}])
@pytest.mark.parametrize("use_distance_map", [True, False])
def test_every_force_kernel_instantiation_reproduces_the_oracle(hip, oracle, monkeypatch, env, use_distance_map):
    """The by-size rule (pedoni_hip.hip plan_force / group_by_size) picks ONE instantiation of the force
    kernel per crowd size -- 2 or 4 lanes per agent for small crowds, the one-lane kernel in its
    default or 94-SGPR build above -- so a crowd of one size would only ever test one of them.
    Here each is pinned in turn on the same crowd (30 000 agents among 100 walls, 300 of them
    packed into one cell: a lane with hundreds of candidates beside lanes with a dozen) and must
    give the oracle's accelerations and 4 ticks of its state, bit for bit, on both wall paths."""
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    sc = random_obstacle_scenario(120.0, 100)
    field = oracle_field(oracle, sc)
    pos, dest, v0, vel = inject_crowd(field, sc.field.size, 30_000, 4, seed=77)
    rng = np.random.default_rng(5)
    pos[:300] = (np.array([60.3, 61.0]) + rng.uniform(0.0, 1.0, (300, 2))).astype(np.float32)   # one crowded cell
    obstacles = sc.obstacle_array()
    cpu = oracle.OracleModel(sc.field.size, use_distance_map=use_distance_map)
    gpu = hip.HipModel(hip.Options(use_distance_map=use_distance_map), sc.field.size, field.distance_map,
                       field.potential_maps, field.unit, obstacles)
    want_symbol = {"1": "force_kernel_queue<0, 8>", "2": "force_kernel_queue_group<0, ", "4": "force_kernel_queue_group<0, "}
    sym, per_wave = gpu.force_kernel_info(30_000)
    if "PEDONI_FORCE_GROUP" in env:
        assert sym.startswith(want_symbol[env["PEDONI_FORCE_GROUP"]]) and per_wave == 64 // int(env["PEDONI_FORCE_GROUP"])
        if "PEDONI_FORCE_GROUP_SLOTS" in env:
            assert f", {env['PEDONI_FORCE_GROUP_SLOTS']}, " in sym
    else:
        build, slots = env["PEDONI_FORCE_KERNEL"].split(":")
        assert sym == ("force_kernel_queue_s94" if build == "s94" else "force_kernel_queue") + f"<0, {slots}>"
    cpu.spawn_pedestrians(field, pos, dest, v0, vel)
    gpu.append(pos, dest, v0, vel)
    gpu.sort_despawn()
    assert bit_equal(gpu.calc_accelerations(cpu.get_pedestrian_count()), cpu.calc_accelerations(field, obstacles)).all()
    for step in range(4):
        cpu.update_states(field, obstacles)
        gpu.update_states()
        cpu.spawn_pedestrians(field)
        gpu.spawn_pedestrians()
        assert np.array_equal(gpu.neighbor_grid_indices(), cpu.neighbor_grid_indices())
        _assert_state_equal(gpu.download(), cpu.download(), f"{env} step {step}")
    gpu.tick_n(6)                      # (captured pairs replay the pinned kernel too)
    for _ in range(6):
        cpu.spawn_pedestrians(field)
        cpu.update_states(field, obstacles)
    gpu.sort_despawn(); cpu.spawn_pedestrians(field)
    _assert_state_equal(gpu.download(), cpu.download(), f"{env} after tick_n")
    gpu.close()


def test_group_kernel_fast_mode_within_1e5_of_exact(hip, oracle, monkeypatch):
    """PEDONI_MATH_FAST through the lanes-per-agent kernel: within north_star's 1e-5 of the exact mode
    per step from identical state, for every agent (same bar as test_fast_math_mode_within_1e5)."""
    sc = random_obstacle_scenario(120.0, 100)
    field = oracle_field(oracle, sc)
    pos, dest, v0, vel = inject_crowd(field, sc.field.size, 40_000, 4, seed=78)
    worst = 0.0
    for group in ("2", "4"):
        monkeypatch.setenv("PEDONI_FORCE_GROUP", group)
        out = {}
        for mode in (abi.MATH_EXACT, abi.MATH_FAST):
            gpu = hip.HipModel(hip.Options(math_mode=mode), sc.field.size, field.distance_map, field.potential_maps,
                               field.unit, sc.obstacle_array())
            gpu.append(pos, dest, v0, vel)
            gpu.sort_despawn()
            acc = gpu.calc_accelerations(len(pos))
            gpu.update_states()
            out[mode] = (acc, gpu.download())
            gpu.close()
        acc_e, (pe, _, ve, _) = out[abi.MATH_EXACT]
        acc_f, (pf, _, vf, _) = out[abi.MATH_FAST]
        scale = np.maximum(np.linalg.norm(ve, axis=1), np.linalg.norm(acc_e, axis=1) * 0.1) + 1e-30
        with np.errstate(invalid="ignore"):
            err = np.linalg.norm(vf.astype(np.float64) - ve, axis=1) / scale
        err = err[np.isfinite(err)]
        worst = max(worst, float(err.max()))
    assert worst <= 1e-5, worst


def _rect_scenario(width, height, margin=None):
    margin = margin if margin is not None else min(width, height) * 0.15
    sc = scn.Scenario()
    sc.field = scn.FieldConfig((width, height))
    sc.waypoints = [scn.SegmentConfig(((margin, margin), (margin, height - margin))),
                    scn.SegmentConfig(((width - margin, margin), (width - margin, height - margin)))]
    sc.obstacles = [scn.SegmentConfig(((0, 0), (0, height)), 0.2), scn.SegmentConfig(((width, 0), (width, height)), 0.2),
                    scn.SegmentConfig(((0, 0), (width, 0)), 0.2), scn.SegmentConfig(((0, height), (width, height)), 0.2),
                    scn.SegmentConfig(((width * 0.5, height * 0.2), (width * 0.5, height * 0.6)), 0.4)]
    return sc


@pytest.mark.parametrize("width,height,n,grid_unit", [
    (300.0, 12.0, 6000, 1.4),      # a corridor: 9 grid rows, 215 columns
    (12.0, 300.0, 6000, 1.4),      # the same on end: 215 rows of 9 cells (rows != cols both ways)
    (3.0, 3.0, 12, 1.4),           # 3 x 3 cells: every cell touches the border
    (40.0, 40.0, 3000, 7.0),       # cells far larger than the 2 m cutoff: 6 x 6 cells, ~80 agents each
    (40.0, 40.0, 3000, 0.5),       # cells smaller than the cutoff: neighbours beyond the 3 x 3 block are missed, as upstream
    (1.3, 25.0, 40, 1.4),          # ONE column of cells
])
@pytest.mark.parametrize("group", ["1", "2", "4"])
def test_odd_grid_shapes_on_every_lane_layout(hip, oracle, monkeypatch, width, height, n, grid_unit, group):
    """Grid shapes the bench never has -- a single row band, a single column, cells above and below
    the cutoff, a field smaller than a wave -- through the one-lane kernel and both lanes-per-agent
    layouts: cell_start, order and state equal to the oracle's, bit for bit, for 5 ticks."""
    monkeypatch.setenv("PEDONI_FORCE_GROUP", group)
    sc = _rect_scenario(width, height)
    field = oracle_field(oracle, sc)
    if width < 2.0:          # one column of cells: too narrow for inject_crowd's clearance -- a file of agents down the middle
        rng = np.random.default_rng(9)
        pos = np.stack([np.full(n, 0.65) + rng.uniform(-0.05, 0.05, n), np.linspace(3.0, height - 3.0, n)], 1).astype(np.float32)
        dest = (np.arange(n) % 2).astype(np.uint32)
        v0 = np.clip(rng.normal(1.34, 0.26, n), 0.5, 2.2).astype(np.float32)
        vel = np.stack([np.zeros(n), np.where(dest == 1, 0.6, -0.6)], 1).astype(np.float32)
    else:
        pos, dest, v0, vel = inject_crowd(field, sc.field.size, n, 2, seed=int(width * 7 + height), clearance=0.3,
                                          min_potential=0.3)
    cpu = oracle.OracleModel(sc.field.size, neighbor_grid_unit=grid_unit)
    gpu = _make_hip(hip, sc, field, neighbor_grid_unit=grid_unit)
    cpu.spawn_pedestrians(field, pos, dest, v0, vel)
    gpu.append(pos, dest, v0, vel)
    gpu.sort_despawn()
    assert np.array_equal(gpu.neighbor_grid_indices(), cpu.neighbor_grid_indices())
    for step in range(5):
        cpu.update_states(field)
        gpu.update_states()
        cpu.spawn_pedestrians(field)
        gpu.spawn_pedestrians()
        assert np.array_equal(gpu.neighbor_grid_indices(), cpu.neighbor_grid_indices()), step
        _assert_state_equal(gpu.download(), cpu.download(), f"{width}x{height} unit {grid_unit} group {group} step {step}")
    gpu.close()
