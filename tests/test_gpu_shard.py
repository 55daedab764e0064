"""The multi-GPU driver BELOW the C-ABI (pedoni_shard_*, pedoni_amd/csrc/shard.hpp) on one GPU:
G shards of one process on one device, the RCCL transport replaced by device copies
(pedoni_shard_local_group_tick_n) -- same driver code, same kernels, same lists -- must equal
the single model bit for bit, with static bands, with bands balanced by agent count, and while
the bands are re-cut every few ticks on a crowd that is far from uniform.  The RCCL calls
themselves (dlopen'ed ncclCommInitRank / ncclSend / ncclRecv) run with a one-rank group."""
import numpy as np
import pytest

from helpers import bit_equal, inject_crowd, oracle_field
from pedoni_amd import abi
from pedoni_amd import scenario as scn

pytestmark = pytest.mark.gpu
CAP = 4096


def _tall_box(width, height):
    sc = scn.Scenario()
    sc.field = scn.FieldConfig((width, height))
    sc.waypoints = [scn.SegmentConfig(((5, 5), (5, height - 5))),
                    scn.SegmentConfig(((width - 5, 5), (width - 5, height - 5)))]
    sc.obstacles = [scn.SegmentConfig(((0, 0), (0, height)), 0.2),
                    scn.SegmentConfig(((width, 0), (width, height)), 0.2),
                    scn.SegmentConfig(((0, 0), (width, 0)), 0.2),
                    scn.SegmentConfig(((0, height), (width, height)), 0.2),
                    scn.SegmentConfig(((width * 0.4, height * 0.3), (width * 0.6, height * 0.7)), 3.0)]
    return sc


def _lopsided_crowd(field, size, n, seed):
    """Three quarters of the agents in the lower third of the field (a C4-type crowd: bands of
    equal ROWS would be badly imbalanced), with vertical drift so that agents cross band edges."""
    pos, dest, v0, vel = inject_crowd(field, size, n, 2, seed=seed)
    rng = np.random.default_rng(seed)
    squeeze = rng.random(n) < 0.66
    pos[squeeze, 1] = (2.0 + (pos[squeeze, 1] - 0.6) * (size[1] * 0.33 / size[1])).astype(np.float32)
    iy = np.clip((pos[:, 1] / field.unit).astype(int), 0, field.shape[0] - 1)
    ix = np.clip((pos[:, 0] / field.unit).astype(int), 0, field.shape[1] - 1)
    ok = field.distance_map[iy, ix] > 0.5
    pos, dest, v0, vel = pos[ok], dest[ok], v0[ok], vel[ok]
    vel[:, 1] += np.where(np.arange(len(pos)) % 2 == 0, 1.2, -1.2).astype(np.float32)
    return pos, dest, v0, vel


@pytest.mark.parametrize("world,balanced,rebalance_every", [(3, False, 0), (4, True, 0), (4, False, 5), (5, True, 4)])
def test_local_group_equals_single_model_bitwise(hip, oracle, world, balanced, rebalance_every):
    import torch
    sc = _tall_box(70.0, 210.0)
    field = oracle_field(oracle, sc)
    pos, dest, v0, vel = _lopsided_crowd(field, sc.field.size, 60_000, seed=40 + world)

    def make():
        return hip.HipModel(hip.Options(), sc.field.size, field.distance_map, field.potential_maps,
                            field.unit, sc.obstacle_array())

    single = make()
    single.append(pos, dest, v0, vel)
    single.sort_despawn()
    rows, cols = single.neighbor_grid_shape()
    idx = single.neighbor_grid_indices().astype(np.int64)
    row_counts = np.diff(idx[::cols][:rows + 1])            # agents per grid row, off the cell_start prefix
    assert row_counts.sum() == single.get_pedestrian_count()
    bounds = abi.balanced_bounds(row_counts, world) if balanced else [(rows * r) // world for r in range(world + 1)]

    # every band holds only ITS texel rows of the field maps (pedoni_hip_create_rows), with room
    # for the re-cut to move the band by up to `slack` grid rows
    slack = 12 if rebalance_every else 0
    stream = torch.cuda.current_stream().cuda_stream
    models, shards = [], []
    for r in range(world):
        rows_needed = abi.shard_map_rows(bounds[r], bounds[r + 1], slack, 1.4, field.unit, field.shape[0])
        m = hip.HipModel(hip.Options(), sc.field.size, field.distance_map, field.potential_maps, field.unit,
                         sc.obstacle_array(), map_rows=rows_needed)
        assert rows_needed[1] - rows_needed[0] < field.shape[0] * (0.75 if world > 3 else 0.9)
        models.append(m)
        m.set_stream(stream)
        shards.append(abi.Shard(m, r, world, bounds, CAP))
        if rebalance_every:
            shards[-1].set_rebalance(rebalance_every, 3, map_slack_rows=slack)
    band_of = np.searchsorted(np.asarray(bounds[1:-1]),
                              np.trunc(pos[:, 1] / np.float32(1.4)).astype(np.int64), side="right")
    for r, (m, s) in enumerate(zip(models, shards)):
        sel = band_of == r
        if sel.any():
            m.append(pos[sel], dest[sel], v0[sel], vel[sel])
        s.begin()
    loads0 = np.array([s.owned_count() for s in shards])
    assert loads0.sum() == single.get_pedestrian_count()
    if balanced:
        assert loads0.max() < 1.25 * loads0.mean(), f"balanced cut is not balanced: {loads0}"
    else:
        assert loads0.max() > 1.5 * loads0.mean()            # the crowd IS lopsided

    # single model:  sort (update sort)^T.   bands:  sort (exchange sort update pack)^T, i.e. the
    # same state one sort short -- a last exchange + sort (no update) lines them up.
    ticks = 23
    for _ in range(ticks):
        single.update_states()
        single.sort_despawn()
    abi.local_group_tick_n(shards, ticks)
    _half_tick(models, shards)
    torch.cuda.synchronize()

    want = single.download()
    parts = [s.download_owned() for s in shards]
    got = [np.concatenate([p[k] for p in parts]) for k in range(4)]
    assert sum(s.owned_count() for s in shards) == len(want[0]) == len(got[0])
    assert np.array_equal(got[1], want[1])
    for k in (0, 2, 3):
        assert bit_equal(got[k], want[k]).all(), f"array {k} differs between the shards and the single model"
    new_bounds = [shards[r].band()[0] for r in range(world)] + [shards[-1].band()[1]]
    if rebalance_every and not balanced:
        loads1 = np.array([s.owned_count() for s in shards])
        assert new_bounds != list(bounds), "the bands were never re-cut"
        assert loads1.max() / loads1.mean() < loads0.max() / loads0.mean(), (loads0, loads1)
    elif not rebalance_every:
        assert new_bounds == list(bounds)
    for s in shards:
        s.close()
    for m in models + [single]:
        m.close()


def _half_tick(models, shards):
    """exchange + unpack + sort/despawn (no update) for every band, through the public halo
    entry points: hands each band its neighbours' freshly packed lists."""
    import torch
    world = len(shards)
    cap = CAP
    words = abi.HipModel.halo_bytes(cap) // 4
    sends = [torch.zeros(words, dtype=torch.int32, device="cuda") for _ in range(world)]
    for m, snd in zip(models, sends):
        m.halo_pack(snd.data_ptr(), cap)
    for r, m in enumerate(models):
        m.halo_unpack(sends[r - 1].data_ptr() if r > 0 else None,
                      sends[r + 1].data_ptr() if r + 1 < world else None, cap)
        m.sort_despawn()


def test_balanced_bounds_cut_by_agents_not_rows(hip):
    counts = np.array([1000] * 10 + [10] * 90, np.uint32)
    b = abi.balanced_bounds(counts, 4, min_rows=2)
    assert b[0] == 0 and b[-1] == 100 and all(b[i + 1] - b[i] >= 2 for i in range(4))
    loads = [int(counts[b[i]:b[i + 1]].sum()) for i in range(4)]
    assert max(loads) <= 1.3 * sum(loads) / 4, loads
    with pytest.raises(abi.PedoniError, match="world"):
        abi.balanced_bounds(counts, 60, min_rows=2)


def test_one_rank_rccl_group_runs_the_direct_calls(hip, oracle):
    """ncclGetUniqueId / ncclCommInitRank / ncclSend / ncclRecv through the dlopen'ed librccl,
    with the one rank this box has: the communicator comes up, a self-addressed token arrives,
    and shard ticks equal the unsharded model bit for bit."""
    sc = _tall_box(60.0, 80.0)
    field = oracle_field(oracle, sc)
    pos, dest, v0, vel = inject_crowd(field, sc.field.size, 20_000, 2, seed=12)

    def make():
        return hip.HipModel(hip.Options(), sc.field.size, field.distance_map, field.potential_maps,
                            field.unit, sc.obstacle_array())
    single, m = make(), make()
    rows, _ = single.neighbor_grid_shape()
    uid = abi.shard_unique_id()
    assert len(uid) == 128 and any(uid)
    s = abi.Shard(m, 0, 1, [0, rows], 2048, unique_id=uid)
    s.selftest()
    single.append(pos, dest, v0, vel)
    m.append(pos, dest, v0, vel)
    s.begin()
    s.tick_n(7)
    s.set_overlap(True)          # split tick: edge rows, pack, exchange on its own stream, interior
    s.tick_n(5)
    s.set_overlap(False)         # (settles the exchange under way; its lists serve the next tick)
    s.tick_n(3)
    single.tick_n(15)
    single.sort_despawn(); m.sort_despawn()
    a, b = single.download(), m.download()
    assert s.owned_count() == single.get_pedestrian_count() == len(a[0])
    assert all(bit_equal(x, y).all() for x, y in zip(a, b))
    s.close(); m.close(); single.close()


def test_sampling_outside_the_uploaded_map_rows_is_loud_not_a_fault(hip, oracle):
    """A model that holds only a slice of the field maps never reads outside it: the sample
    returns the out-of-field value and the sticky device status makes the next read of device
    state fail (PEDONI_E_CAPACITY), instead of an out-of-bounds access."""
    sc = _tall_box(60.0, 120.0)
    field = oracle_field(oracle, sc)
    pos, dest, v0, vel = inject_crowd(field, sc.field.size, 4000, 2, seed=3)
    lower = pos[:, 1] < 50.0
    m = hip.HipModel(hip.Options(), sc.field.size, field.distance_map, field.potential_maps, field.unit,
                     sc.obstacle_array(), map_rows=(0, 240))          # texel rows of y < 60 m only
    m.append(pos[lower], dest[lower], v0[lower], vel[lower])
    m.tick_n(3)
    assert m.get_pedestrian_count() > 0                               # inside the slice: fine
    m.append(pos[~lower][:10], dest[~lower][:10], v0[~lower][:10], vel[~lower][:10])
    m.tick_n(1)
    with pytest.raises(abi.PedoniError, match="field-map row"):
        m.get_pedestrian_count()
    m.close()
    with pytest.raises(abi.PedoniError, match="map row range"):
        hip.HipModel(hip.Options(), sc.field.size, field.distance_map, field.potential_maps, field.unit,
                     sc.obstacle_array(), map_rows=(10, 10))
