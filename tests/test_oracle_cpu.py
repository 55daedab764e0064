"""CPU suite, part 1: the oracle itself.

Pins the oracle against every asserted value the reference's own tests hold (6 values:
util.rs:148-163), against an independently written NumPy-float32 restatement, and
against closed-form cases.  Everything beyond the 6 values is "parity unpinned"
(no Rust toolchain here to run the reference).
"""
import ctypes
import ctypes.util

import numpy as np
import pytest

import np_ref
from helpers import GOLDEN, bit_equal, box_scenario, inject_crowd, oracle_field, \
    random_obstacle_scenario
from pedoni_amd import scenario as scn


# ---- the reference's own known-answer tests -----------------------------------------------
def test_reference_kat_bilinear(oracle):
    """util.rs:157-163 test_bilinear (assert_float_absolute_eq, eps 1e-6)."""
    grid = np.array([[1.0, 0.0, 4.0], [3.0, 1.0, -1.0]], np.float32)
    for (x, y), want in (((0.0, 0.0), 1.0), ((0.5, 0.0), 0.5), ((0.0, 0.25), 1.5), ((0.5, 0.5), 1.25)):
        assert abs(oracle.bilinear(grid, x, y) - want) <= 1e-6
        assert abs(float(np_ref.bilinear(grid, x, y)) - want) <= 1e-6


def test_reference_kat_distance_from_line(oracle):
    """util.rs:149-154 test_distance_from_line."""
    line = [[1.0, 1.0], [4.0, 1.0]]
    assert abs(np.linalg.norm(oracle.distance_from_line((2.0, 3.0), line)) - 2.0) <= 1e-6
    assert abs(np.linalg.norm(oracle.distance_from_line((0.0, 0.25), line)) - 1.25) <= 1e-6


def test_distance_from_line_degenerate_quirk(oracle):
    """util.rs:97-98: a zero-length segment returns a - line[0] (i.e. point - 2*line[0])."""
    d = oracle.distance_from_line((5.0, 7.0), [[2.0, 3.0], [2.0, 3.0]])
    assert np.array_equal(d, np.array([5 - 2 - 2, 7 - 3 - 3], np.float32))


def test_bilinear_out_of_bounds_is_1e12(oracle):
    grid = np.ones((4, 5), np.float32)
    assert oracle.bilinear(grid, -3.0, 1.0) == np.float32(1e12)
    assert oracle.bilinear(grid, 2.0, 17.0) == np.float32(1e12)
    assert np.isnan(oracle.bilinear(grid, np.nan, 1.0))


def test_sobel_sign_and_scale(oracle):
    """util.rs:71-74 returns left-right / up-down: -8 * gradient of a linear ramp."""
    yy, xx = np.mgrid[0:16, 0:16].astype(np.float32)
    ramp = (3 * xx + 5 * yy).astype(np.float32)
    g = oracle.sobel_filter(ramp, 7.3, 6.1)
    assert np.allclose(g, [-8 * 3, -8 * 5], rtol=1e-5)


# ---- expf restated (what the HIP kernels evaluate) ------------------------------------------
def _libm_expf(x):
    libm = ctypes.CDLL(ctypes.util.find_library("m") or "libm.so.6")
    libm.expf.restype = ctypes.c_float
    libm.expf.argtypes = [ctypes.c_float]
    return np.array([libm.expf(float(v)) for v in x], np.float32)


def test_expf_restatement_equals_host_libm(oracle):
    """The f64 replay of glibc's expf is bit-identical to this host's libm on the force
    path's argument range (x <= 0).  An exhaustive sweep of all 2.2e9 floats in
    [-104, 88] found two 1-ulp exceptions (x = -63.0994606 -> 1.7e-28, and one x > 0,
    unreachable: every argument on the path is -b/0.3 or -d/0.2)."""
    rng = np.random.default_rng(0)
    x = np.concatenate([-rng.uniform(0, 40, 150000), -rng.lognormal(0, 2.5, 50000),
                        np.linspace(-104.5, 0, 20001),
                        [0.0, -0.0, -1e-30, -87.99, -88.0, -88.01, -103.9, -103.98, -104.0, -1e9,
                         -np.inf, np.nan]]).astype(np.float32)
    got, want = oracle.expf_restated(x), _libm_expf(x)
    eq = bit_equal(got, want)
    assert eq.all(), f"mismatch at {x[~eq][:8]}"


# ---- oracle vs the independent NumPy restatement ----------------------------------------------
@pytest.mark.parametrize("seed,n", [(1, 120), (2, 300)])
def test_oracle_matches_numpy_restatement(oracle, seed, n):
    sc = random_obstacle_scenario(40.0, 12, seed=seed)
    field = oracle_field(oracle, sc)
    pos, dest, v0, vel = inject_crowd(field, sc.field.size, n, 4, seed=seed)
    pos[3] = [-2.0, 5.0]          # out of grid -> vanishes
    pos[4] = [np.nan, 5.0]        # NaN -> vanishes
    m = oracle.OracleModel(sc.field.size)
    m.spawn_pedestrians(field, pos, dest, v0, vel)
    shape = oracle.neighbor_grid_shape(sc.field.size, 1.4)
    rp, rd, rv, r0, rstart = np_ref.sort_despawn(pos, dest, vel, v0, field.unit, field.potential_maps,
                                                 1.4, shape)
    op, od, ov, o0 = m.download()
    assert np.array_equal(m.neighbor_grid_indices(), rstart)
    assert np.array_equal(od, rd) and bit_equal(op, rp).all() and bit_equal(ov, rv).all()

    racc = np_ref.accelerations(rp, rd, rv, r0, rstart, field.unit, field.distance_map,
                                field.potential_maps, 1.4, shape)
    oacc = m.calc_accelerations(field)
    assert bit_equal(oacc, racc).all(), f"{np.count_nonzero(~bit_equal(oacc, racc))} acc values differ"

    m.update_states(field)
    np_pos, np_vel = np_ref.integrate(rp, rv, r0, racc)
    op, od, ov, o0 = m.download()
    assert bit_equal(op, np_pos).all() and bit_equal(ov, np_vel).all()


# ---- closed-form cases ------------------------------------------------------------------------
def test_two_body_force_closed_form(oracle):
    """Two agents at rest 1 m apart along x in an open field: b = d/2... wait, with zero
    neighbour velocity t1 = d, b = sqrt((2d)^2)/2 = d, grad b = n, so
    |f| = (2.1/0.3) exp(-d/0.3), directed away from the neighbour, halved when the
    neighbour is behind (outside the 200-degree field of view)."""
    sc = box_scenario(60.0)
    field = oracle_field(oracle, sc)
    pos = np.array([[30.0, 30.0], [31.0, 30.0]], np.float32)
    dest = np.array([1, 1], np.uint32)     # both walk towards +x
    v0 = np.array([1.3, 1.3], np.float32)
    vel = np.zeros((2, 2), np.float32)
    m = oracle.OracleModel(sc.field.size)
    m.spawn_pedestrians(field, pos, dest, v0, vel)
    acc = m.calc_accelerations(field)
    # remove goal and wall terms by differencing against single-agent runs
    solo = []
    for k in range(2):
        s = oracle.OracleModel(sc.field.size)
        s.spawn_pedestrians(field, pos[k:k + 1], dest[k:k + 1], v0[k:k + 1], vel[k:k + 1])
        solo.append(s.calc_accelerations(field)[0])
    pair = acc - np.array(solo)
    mag = (2.1 / 0.3) * np.exp(-1.0 / 0.3)
    # agent 0 sees agent 1 ahead: full force, pushing it back (-x)
    assert np.allclose(pair[0], [-mag, 0.0], rtol=2e-5, atol=1e-6)
    # agent 1 has agent 0 behind it: halved force, pushing it forward (+x)
    assert np.allclose(pair[1], [0.5 * mag, 0.0], rtol=2e-5, atol=1e-6)


def test_goal_relaxation_and_speed_clamp(oracle):
    """A lone agent: acc = (e*v0 - v)/0.5 + wall term; |v| never exceeds 1.3*v0."""
    sc = box_scenario(60.0)
    field = oracle_field(oracle, sc)
    m = oracle.OracleModel(sc.field.size)
    m.spawn_pedestrians(field, np.array([[30.0, 30.0]], np.float32), np.array([1], np.uint32),
                        np.array([1.0], np.float32), np.array([[0.0, 5.0]], np.float32))
    m.update_states(field)
    _, _, vel, _ = m.download()
    assert abs(np.linalg.norm(vel[0]) - 1.3) < 1e-5
    for _ in range(30):
        m.spawn_pedestrians(field)
        m.update_states(field)
    _, _, vel, _ = m.download()
    assert np.allclose(vel[0], [1.0, 0.0], atol=2e-2)   # relaxed to e * v0, e = +x


def test_pairs_two_cells_apart_are_missed_by_design(oracle):
    """grid unit 1.4 < cutoff 2.0, so agents 1.5-2.0 m apart but two cells away
    exert no force in grid mode, while brute-force mode sees them."""
    sc = box_scenario(60.0)
    field = oracle_field(oracle, sc)
    pos = np.array([[28.05 + 1.35, 30.0], [28.05 + 1.35 + 1.5, 30.0]], np.float32)  # cells 21 and 22..23
    pos = np.array([[29.35, 30.0], [31.2, 30.0]], np.float32)   # cells 20 and 22, 1.85 m apart
    dest = np.array([1, 1], np.uint32)
    v0 = np.array([1.3, 1.3], np.float32)
    vel = np.zeros((2, 2), np.float32)
    acc = {}
    for grid in (True, False):
        m = oracle.OracleModel(sc.field.size, use_neighbor_grid=grid)
        m.spawn_pedestrians(field, pos, dest, v0, vel)
        acc[grid] = m.calc_accelerations(field)
    assert not np.array_equal(acc[True], acc[False])
    s = oracle.OracleModel(sc.field.size)
    s.spawn_pedestrians(field, pos[:1], dest[:1], v0[:1], vel[:1])
    assert bit_equal(acc[True][0], s.calc_accelerations(field)[0]).all()


# ---- fast marching / field builder -------------------------------------------------------------
def test_fmm_distance_map_is_near_euclidean(oracle):
    sc = box_scenario(30.0)
    field = oracle_field(oracle, sc)
    # centre of the box: nearest wall 15 m away; first-order FMM over-estimates slightly
    d = field.get_obstacle_distance((15.0, 15.0))
    assert 14.5 < d < 16.5
    # distance grows monotonically from a wall along the mid-line
    vals = [field.get_obstacle_distance((x, 15.0)) for x in np.arange(1.0, 14.0, 1.0)]
    assert all(b > a for a, b in zip(vals, vals[1:]))


def test_fmm_potential_prefers_going_around_walls(oracle):
    sc = scn.load(GOLDEN / "scenarios" / "narrow_gap.toml")
    field = oracle_field(oracle, sc)
    # waypoint 1 (x = 12) lies behind the wall at x = 10; the gap is y in (10, 13)
    through_gap = field.get_potential(1, (8.0, 11.5))
    behind_wall = field.get_potential(1, (8.0, 5.0))
    assert through_gap < behind_wall < 1e5
    assert field.obstacle_exist[0].all() and field.obstacle_exist[:, 0].all()


def test_fmm_heap_tie_order_is_total(oracle):
    """A symmetric seed gives a symmetric result: the pop order is defined by a total
    order, so the result does not depend on heap internals."""
    pot = np.full((21, 21), np.finfo(np.float32).max, np.float32)
    pot[10, 10] = 0.0
    out = oracle.apply_fmm(pot, np.full((21, 21), 0.25, np.float32))
    assert np.allclose(out, out[::-1, :]) and np.allclose(out, out[:, ::-1])
    assert np.allclose(out, out.T)
    assert out[10, 14] == np.float32(1.0)
