"""The C++ host mirror of pedoni-simulator (Simulator::new / tick / list_pedestrians,
lib.rs:27-105) driving the HIP backend, against the oracle fed with the same build-owned
RNG streams (spawn positions: options.seed; desired speeds: options.seed ^ 0x5eed)."""
import numpy as np
import pytest

from helpers import GOLDEN, bit_equal, oracle_field
from pedoni_amd import scenario as scn

pytestmark = pytest.mark.gpu


def test_simulator_narrow_gap_matches_oracle(hip, oracle):
    from pedoni_amd import host
    text = (GOLDEN / "scenarios" / "narrow_gap.toml").read_text()
    seed = 2024
    sim = host.Simulator(host.SimulatorOptions(seed=seed), host.Scenario(text))

    sc = scn.loads(text)
    field = oracle_field(oracle, sc)
    # the product's own field builder produced the same maps (also checked on CPU)
    assert np.array_equal(sim.field.distance_map.view(np.uint32), field.distance_map.view(np.uint32))

    rng = oracle.Rng(seed)                                       # Simulator::new, lib.rs:37-52
    p1, p2 = (np.array(p, np.float32) for p in sc.waypoints[0].line)
    u = np.array([rng.f32() for _ in range(50)], np.float32)
    pos = (p1[None] * (np.float32(1) - u)[:, None] + p2[None] * u[:, None]).astype(np.float32)
    cpu = oracle.OracleModel(sc.field.size, seed=seed ^ 0x5eed)
    cpu.spawn_pedestrians(field, pos, np.ones(50, np.uint32))   # v0 drawn inside the model

    assert sim.step == 0
    counts = []
    for t in range(250):
        m = sim.tick()                                           # lib.rs:64-100
        cpu.spawn_pedestrians(field)
        cpu.update_states(field)
        wp, wd, _, _ = cpu.download()
        got = sim.list_pedestrians()
        assert m["active_ped_count"] == len(wp) == len(got), f"tick {t}"
        assert bit_equal(np.stack([got["x"], got["y"]], 1), wp).all(), f"tick {t}"
        assert np.array_equal(got["destination"], wd)
        assert m["time_spawn"] >= 0 and m["time_calc_state"] > 0
        counts.append(len(got))
    assert sim.step == 250 and counts[0] == 50 and counts[-1] < 50
    sim.close()


def test_simulator_periodic_spawns_match_oracle_replica(hip, oracle):
    """Simulator::tick with periodic spawners (lib.rs:67-85): count = poisson(frequency / 10),
    pos = lerp(origin line, f32()) from the seeded stream, then spawn_pedestrians (append +
    sort/despawn: the general sort form runs on every tick that spawns) and update_states.
    The replica drives the oracle with the same streams; positions must match bit for bit."""
    from pedoni_amd import host
    text = """
[field]
size = [60, 40]
[[waypoints]]
line = [[5, 5], [5, 35]]
[[waypoints]]
line = [[55, 5], [55, 35]]
[[obstacles]]
line = [[30, 0], [30, 15]]
width = 1
[[obstacles]]
line = [[30, 40], [30, 22]]
width = 1.5
[[pedestrians]]
origin = 0
destination = 1
spawn = { kind = "periodic", frequency = 40.0 }
[[pedestrians]]
origin = 1
destination = 0
spawn = { kind = "periodic", frequency = 25.0 }
[[pedestrians]]
origin = 0
destination = 1
spawn = { kind = "once", count = 7 }
"""
    seed = 99
    sim = host.Simulator(host.SimulatorOptions(seed=seed), host.Scenario(text))
    sc = scn.loads(text)
    field = oracle_field(oracle, sc)
    rng = oracle.Rng(seed)
    cpu = oracle.OracleModel(sc.field.size, seed=seed ^ 0x5eed)

    def lerp_line(w, u):
        p1, p2 = (np.array(p, np.float32) for p in sc.waypoints[w].line)
        return p1 * (np.float32(1) - u) + p2 * u

    new = [lerp_line(0, np.float32(rng.f32())) for _ in range(7)]            # lib.rs:37-52
    cpu.spawn_pedestrians(field, np.array(new, np.float32), np.ones(7, np.uint32))
    for t in range(120):
        m = sim.tick()
        pos, dest = [], []
        for p in sc.pedestrians:                                              # lib.rs:70-84
            if isinstance(p.spawn, scn.SpawnPeriodic):
                for _ in range(rng.poisson(p.spawn.frequency / 10.0)):
                    pos.append(lerp_line(p.origin, np.float32(rng.f32())))
                    dest.append(p.destination)
        cpu.spawn_pedestrians(field, np.array(pos, np.float32).reshape(-1, 2), np.array(dest, np.uint32))
        cpu.update_states(field)
        wp, wd, _, _ = cpu.download()
        got = sim.list_pedestrians()
        assert m["active_ped_count"] == len(wp) == len(got), f"tick {t}"
        assert np.array_equal(got["destination"], wd), f"tick {t}"
        assert bit_equal(np.stack([got["x"], got["y"]], 1), wp).all(), f"tick {t}"
    assert len(got) > 300
    sim.close()


def test_simulator_periodic_spawns_grow_the_crowd(hip):
    from pedoni_amd import host
    text = """
[field]
size = [60, 40]
[[waypoints]]
line = [[5, 5], [5, 35]]
[[waypoints]]
line = [[55, 5], [55, 35]]
[[obstacles]]
line = [[30, 0], [30, 15]]
width = 1
[[pedestrians]]
origin = 0
destination = 1
spawn = { kind = "periodic", frequency = 40.0 }
[[pedestrians]]
origin = 1
destination = 0
spawn = { kind = "periodic", frequency = 25.0 }
"""
    sim = host.Simulator(host.SimulatorOptions(seed=7), host.Scenario(text))
    n = [sim.tick()["active_ped_count"] for _ in range(60)]
    assert n[-1] > n[5] > 0                      # Poisson(4) + Poisson(2.5) arrivals per tick
    peds = sim.list_pedestrians()
    assert set(peds["destination"]) <= {0, 1} and np.isfinite(peds["x"]).all()
    sim.close()


SPAWN_SCENARIO = """
[field]
size = [80, 50]
[[waypoints]]
line = [[5, 5], [5, 45]]
[[waypoints]]
line = [[75, 5], [75, 45]]
[[waypoints]]
line = [[30, 48], [50, 48]]
[[obstacles]]
line = [[40, 0], [40, 20]]
width = 1
[[pedestrians]]
origin = 0
destination = 1
spawn = { kind = "periodic", frequency = 60.0 }
[[pedestrians]]
origin = 1
destination = 0
spawn = { kind = "periodic", frequency = 35.0 }
[[pedestrians]]
origin = 2
destination = 0
spawn = { kind = "periodic", frequency = 0.5 }
[[pedestrians]]
origin = 0
destination = 2
spawn = { kind = "once", count = 11 }
"""


def _snapshot(sim):
    p = sim.list_pedestrians()
    return np.stack([p["x"], p["y"]], 1), p["destination"].copy()


def test_device_spawning_equals_host_spawning(hip):
    """Simulator::tick_n evaluates the periodic spawners on the device (spawn_kernel replays
    the position and desired-speed streams draw for draw): after the same number of ticks
    the crowd is bit-identical to per-tick host spawning, also when the two are mixed."""
    from pedoni_amd import host
    sc = host.Scenario(SPAWN_SCENARIO)
    a = host.Simulator(host.SimulatorOptions(seed=5), sc)       # host spawns, one tick at a time
    b = host.Simulator(host.SimulatorOptions(seed=5), sc)       # device spawns, batches
    c = host.Simulator(host.SimulatorOptions(seed=5), sc)       # mixed
    for _ in range(180):
        a.tick()
    for n in (1, 59, 120):
        m = b.tick_n(n)
    for n, host_ticks in ((40, 7), (3, 30), (100, 0)):
        c.tick_n(n)
        for _ in range(host_ticks):
            c.tick()
    assert a.step == b.step == c.step == 180
    pa, da = _snapshot(a)
    assert len(pa) > 500 and m["active_ped_count"] == len(pa)
    for other in (b, c):
        po, do = _snapshot(other)
        assert len(po) == len(pa) and np.array_equal(do, da)
        assert bit_equal(po, pa).all()
    for s in (a, b, c):
        s.close()


def test_checkpoint_resume_continues_bit_for_bit(hip, tmp_path):
    """Build-owned checkpoint (upstream has none and list_pedestrians drops velocity
    and desired speed): step counter, both generator states, full SoA state.  A run resumed
    from the file -- in a fresh Simulator -- must continue exactly like the uninterrupted
    one, Poisson arrivals and desired-speed draws included, whether the checkpoint is taken
    after host ticks or after a device-spawning batch."""
    from pedoni_amd import abi, host
    sc = host.Scenario(SPAWN_SCENARIO)
    opt = host.SimulatorOptions(seed=11)
    ref = host.Simulator(opt, sc)
    for _ in range(90):
        ref.tick()
    want_pos, want_dest = _snapshot(ref)

    a = host.Simulator(opt, sc)
    for _ in range(25):
        a.tick()
    a.save_checkpoint(tmp_path / "t25.ckpt")
    a.close()
    b = host.Simulator.resume(opt, sc, tmp_path / "t25.ckpt")
    assert b.step == 25
    b.tick_n(35)                                   # device-side spawning after the resume
    b.save_checkpoint(tmp_path / "t60.ckpt")       # ... and a checkpoint taken right after it
    b.close()
    c = host.Simulator.resume(opt, sc, tmp_path / "t60.ckpt")
    assert c.step == 60
    for _ in range(30):
        c.tick()
    got_pos, got_dest = _snapshot(c)
    assert c.step == ref.step == 90
    assert len(got_pos) == len(want_pos) > 300 and np.array_equal(got_dest, want_dest)
    assert bit_equal(got_pos, want_pos).all()
    c.close()
    ref.close()

    # the world must match: other options or another scenario are refused
    with pytest.raises(abi.PedoniError, match="another scenario or other simulator options"):
        host.Simulator.resume(host.SimulatorOptions(seed=11, neighbor_grid_unit=2.0), sc, tmp_path / "t60.ckpt")
    (tmp_path / "junk.ckpt").write_bytes(b"not a checkpoint at all")
    with pytest.raises(abi.PedoniError, match="not a pedoni checkpoint"):
        host.Simulator.resume(opt, sc, tmp_path / "junk.ckpt")
    blob = (tmp_path / "t60.ckpt").read_bytes()
    (tmp_path / "cut.ckpt").write_bytes(blob[: len(blob) // 2])
    with pytest.raises(abi.PedoniError, match="truncated"):
        host.Simulator.resume(opt, sc, tmp_path / "cut.ckpt")


def test_tick_n_without_neighbor_grid_spawns_on_the_host(hip):
    """Device spawning is a neighbor-grid feature; with `use_neighbor_grid = false` (the
    reference's brute-force option, sfm.rs:78-88,157-185) tick_n must fall back to per-tick
    host spawning and still equal tick() x n."""
    from pedoni_amd import host
    sc = host.Scenario(SPAWN_SCENARIO)
    opt = host.SimulatorOptions(seed=21, use_neighbor_grid=False)
    a, b = host.Simulator(opt, sc), host.Simulator(opt, sc)
    for _ in range(40):
        a.tick()
    m = b.tick_n(40)
    pa, da = _snapshot(a)
    pb, db = _snapshot(b)
    assert a.step == b.step == 40 and m["active_ped_count"] == len(pa) == len(pb) > 100
    assert np.array_equal(da, db) and bit_equal(pa, pb).all()
    a.close()
    b.close()
