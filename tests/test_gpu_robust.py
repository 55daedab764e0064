"""Failure modes and launch forms that must never be silent: the device status word, halo
lists that outgrow the host's bound of the arrays, pending appends at an exchange, and the
captured-graph form of tick_n against the eager launches."""
import numpy as np
import pytest

from helpers import bit_equal, inject_crowd, oracle_field, random_obstacle_scenario
from pedoni_amd import abi
from pedoni_amd import scenario as scn

pytestmark = pytest.mark.gpu


def _model(hip, oracle, n=5000, L=80.0, seed=11, diagnostics=False, **opt):
    sc = random_obstacle_scenario(L, 40, seed=seed)
    field = oracle_field(oracle, sc)
    pos, dest, v0, vel = inject_crowd(field, sc.field.size, n, 4, seed=seed)
    gpu = hip.HipModel(hip.Options(**opt), sc.field.size, field.distance_map, field.potential_maps,
                       field.unit, sc.obstacle_array(), diagnostics=diagnostics)
    gpu.append(pos, dest, v0, vel)
    return sc, field, gpu, (pos, dest, v0, vel)


@pytest.mark.parametrize("word,match", [(1, "row count"), (2, "more live agents"), (64, "status word")])
def test_device_status_word_fails_every_read_of_device_state(hip, oracle, word, match):
    """kernels.hpp STATUS_*: the scan's integrity check and the live-count bound raise a sticky
    device word; get_pedestrian_count / download / list_pedestrians / owned_count must all fail
    while it is set (never continue on a wrong cell_start).  The hook that sets
    the word exists in the diagnostics build of the library only (libpedoni_hip_diag.so)."""
    _, _, gpu, _ = _model(hip, oracle, diagnostics=True)
    gpu.tick_n(3)
    n = gpu.get_pedestrian_count()
    assert n > 0
    gpu.debug_set_status(word)
    for call in (gpu.get_pedestrian_count, gpu.download, gpu.list_pedestrians, gpu.owned_count):
        with pytest.raises(abi.PedoniError, match=match):
            call()
    gpu.tick_n(2)                                   # sticky: ticking does not clear it
    with pytest.raises(abi.PedoniError, match=match):
        gpu.get_pedestrian_count()
    gpu.debug_set_status(0)
    assert gpu.get_pedestrian_count() <= n
    gpu.close()


def test_scan_integrity_holds_on_a_real_run(hip, oracle):
    """The check itself: thousands of ticks' worth of row / cell counts agree (status stays 0
    through despawns, general-form passes and appends)."""
    sc, field, gpu, (pos, dest, v0, vel) = _model(hip, oracle, n=20_000, L=120.0, seed=5)
    for k in range(6):
        gpu.tick_n(25)
        gpu.append(pos[k * 100:(k + 1) * 100], dest[k * 100:(k + 1) * 100])   # forces a general pass
        assert gpu.get_pedestrian_count() > 0       # raises if the status word were set
    gpu.close()


def test_graph_replay_equals_eager_launches(hip, oracle):
    """pedoni_hip_tick_n replays captured tick pairs in steady state; explicit
    sort_despawn / update_states calls never do.  Same kernels, same bits -- also across an
    odd step count, a re-capture after the host bound tightens, and an append in between."""
    _, _, a, crowd = _model(hip, oracle, n=30_000, L=130.0, seed=21)
    _, _, b, _ = _model(hip, oracle, n=30_000, L=130.0, seed=21)
    pos, dest, v0, vel = crowd

    def eager(m, steps):
        for _ in range(steps):
            m.sort_despawn()
            m.update_states()

    for steps in (1, 7, 20, 5):
        a.tick_n(steps)
        eager(b, steps)
        assert a.get_pedestrian_count() == b.get_pedestrian_count()   # tightens the bound -> re-capture
        ga, gb = a.download(), b.download()
        assert all(bit_equal(x, y).all() for x, y in zip(ga, gb))
    a.append(pos[:50] + 0.01, dest[:50], v0[:50], vel[:50])
    b.append(pos[:50] + 0.01, dest[:50], v0[:50], vel[:50])
    a.tick_n(9)
    eager(b, 9)
    a.sort_despawn(); b.sort_despawn()
    ga, gb = a.download(), b.download()
    assert all(bit_equal(x, y).all() for x, y in zip(ga, gb))
    assert np.array_equal(a.neighbor_grid_indices(), b.neighbor_grid_indices())
    a.close(); b.close()


@pytest.mark.parametrize("every", [1, 2, 3, 5, 9, 17])
def test_event_timed_ticks_between_replayed_pairs(hip, oracle, every):
    """bench.py times the force kernel on every n-th tick only (those launch eagerly, the others replay
    captured runs of 16 / 8 / 4 / 2 ticks, whichever fits the stretch between two timed ticks and the end
    of the call).  With n odd the runs start on both halves of the ping-pong buffers: one captured graph
    per length and half.  Same bits as plain eager launches, and the timed launches are counted."""
    _, _, a, _ = _model(hip, oracle, n=30_000, L=130.0, seed=23)
    _, _, b, _ = _model(hip, oracle, n=30_000, L=130.0, seed=23)
    a.tick_n(4); a.synchronize()
    for _ in range(4):
        b.sort_despawn(); b.update_states()
    a.profile(True, kernels=[abi.K_FORCE], every=every)
    a.kernel_times(reset=True)
    for steps in (20, 7, 41):
        a.tick_n(steps)
        for _ in range(steps):
            b.sort_despawn(); b.update_states()
    times = a.kernel_times(reset=True)
    a.profile(False)
    assert times["force_integrate"]["launches"] == len([t for t in range(4, 72) if t % every == 0])
    a.tick_n(3)
    for _ in range(3):
        b.sort_despawn(); b.update_states()
    ga, gb = a.download(), b.download()
    assert all(bit_equal(x, y).all() for x, y in zip(ga, gb))
    a.close(); b.close()


def _tall_box(width, height):
    sc = scn.Scenario()
    sc.field = scn.FieldConfig((width, height))
    sc.waypoints = [scn.SegmentConfig(((5, 5), (5, height - 5))),
                    scn.SegmentConfig(((width - 5, 5), (width - 5, height - 5)))]
    sc.obstacles = [scn.SegmentConfig(((0, 0), (0, height)), 0.2),
                    scn.SegmentConfig(((width, 0), (width, height)), 0.2),
                    scn.SegmentConfig(((0, 0), (width, 0)), 0.2),
                    scn.SegmentConfig(((0, height), (width, height)), 0.2)]
    return sc


def test_interior_band_with_tight_halo_capacity(hip, oracle):
    """an interior band receives n_below + n_above agents, up to twice the
    list capacity; on the first tick after load() nothing is dropped yet, so the live count
    exceeds the old host bound (+ one capacity).  With the capacity only slightly above one
    list the three-band run must still equal the single model bit for bit -- and the device
    must not have flagged an overflow."""
    import torch
    from pedoni_amd.sharded import ShardedModel

    world, n = 3, 60_000
    sc = _tall_box(70.0, 126.0)
    field = oracle_field(oracle, sc)
    pos, dest, v0, vel = inject_crowd(field, sc.field.size, n, 2, seed=9)
    vel[:, 1] += np.where(np.arange(n) % 2 == 0, 1.2, -1.2).astype(np.float32)

    def make():
        return hip.HipModel(hip.Options(), sc.field.size, field.distance_map, field.potential_maps,
                            field.unit, sc.obstacle_array())

    single = make()
    single.append(pos, dest, v0, vel)
    single.sort_despawn()
    # one list = the owned agents of two boundary rows: 2 x 70 m x 1.4 m x 6.8 /m^2 ~ 1330
    rows_per_list = 2 * 70.0 * 1.4 * n / (70.0 * 126.0)
    cap = int(rows_per_list * 1.25)
    assert 2 * rows_per_list > cap + 200            # both lists together exceed one capacity
    stream = torch.cuda.current_stream().cuda_stream
    words = hip.HipModel.halo_bytes(cap) // 4
    sends = [torch.zeros(words, dtype=torch.int32, device="cuda") for _ in range(world)]
    models, bands = [make() for _ in range(world)], []
    for r, m in enumerate(models):
        m.set_stream(stream)
        bands.append(ShardedModel(m, r, world, halo_cap=cap, gather=lambda s, rv: None,
                                  send=sends[r], recv=sends))
    owner = bands[0].owner_of(pos[:, 1])
    for r, b in enumerate(bands):
        b.load(pos[owner == r], dest[owner == r], v0[owner == r], vel[owner == r])
    ticks = 20                                      # crosses two tighten intervals
    for _ in range(ticks):
        single.update_states()
        single.sort_despawn()
    for t in range(ticks + 1):
        for b in bands:
            b.pack()
        for b in bands:
            b.unpack()
            b.model.sort_despawn()
            if t < ticks:
                b.model.update_states()
    torch.cuda.synchronize()
    want = single.download()
    parts = [b.download_owned() for b in bands]     # raises if a band flagged an overflow
    got = [np.concatenate([p[k] for p in parts]) for k in range(4)]
    assert sum(b.owned_count() for b in bands) == len(want[0]) == len(got[0])
    assert np.array_equal(got[1], want[1])
    for k in (0, 2, 3):
        assert bit_equal(got[k], want[k]).all()
    for m in models + [single]:
        m.close()


def test_exchange_with_pending_appends_is_rejected(hip, oracle):
    """agents appended by the host since the last pass would be neither own nor
    received once the exchanged lists land behind them -- halo_unpack refuses instead of
    dropping them."""
    import torch
    from pedoni_amd.sharded import ShardedModel
    sc = _tall_box(40.0, 60.0)
    field = oracle_field(oracle, sc)
    pos, dest, v0, vel = inject_crowd(field, sc.field.size, 4000, 2, seed=2)
    m = hip.HipModel(hip.Options(), sc.field.size, field.distance_map, field.potential_maps,
                     field.unit, sc.obstacle_array())
    cap = 1024
    send = torch.zeros(hip.HipModel.halo_bytes(cap) // 4, dtype=torch.int32, device="cuda")
    m.set_stream(torch.cuda.current_stream().cuda_stream)
    band = ShardedModel(m, 0, 1, halo_cap=cap, gather=lambda s, rv: None, send=send, recv=[send])
    band.load(pos, dest, v0, vel)
    band.pack()
    m.append(pos[:10], dest[:10], v0[:10], vel[:10])
    with pytest.raises(abi.PedoniError, match="pending"):
        band.unpack()
    m.sort_despawn()
    band.pack()
    band.unpack()                                   # fine after a pass
    with pytest.raises(abi.PedoniError, match="null send"):
        m.halo_tick(None, None, None, cap)
    m.close()


def test_captured_tick_is_dropped_when_a_baked_in_argument_changes(hip, oracle, monkeypatch):
    """the captured tick pair bakes in the band (kernel arguments), and its key held only
    bounds / buffer indices.  clear + set_band (halo_cap 0) + the same number of agents used to
    replay the OLD band.  Graph run == eager run (PEDONI_NO_GRAPH=1) through that sequence."""
    def run(no_graph):
        if no_graph:
            monkeypatch.setenv("PEDONI_NO_GRAPH", "1")
        else:
            monkeypatch.delenv("PEDONI_NO_GRAPH", raising=False)
        _, _, m, (pos, dest, v0, vel) = _model(hip, oracle, n=20_000, L=100.0, seed=31)
        m.tick_n(6)                                   # steady state: the tick pair is captured (graph run)
        rows, _ = m.neighbor_grid_shape()
        m.clear()
        m.set_band(12, rows - 12, 0)                  # agents outside rows [11, rows - 12] are now dropped
        m.append(pos, dest, v0, vel)                  # same count: the old key would match
        m.tick_n(6)
        out = m.download()
        m.close()
        return out
    g, e = run(False), run(True)
    assert len(g[0]) == len(e[0]) < 20_000
    assert np.array_equal(g[1], e[1]) and all(bit_equal(g[k], e[k]).all() for k in (0, 2, 3))
