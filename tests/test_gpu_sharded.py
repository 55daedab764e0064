"""Row-band sharding on ONE GPU: G bands emulated by G models on the same device, the
all-gather replaced by handing every band the list of all send buffers.  The merged result
must equal the single-model run bit for bit (same kernels, same order)."""
import numpy as np
import pytest

from helpers import bit_equal, box_scenario, inject_crowd, oracle_field, random_obstacle_scenario
from pedoni_amd import scenario as scn

pytestmark = pytest.mark.gpu


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


@pytest.mark.parametrize("world,n", [(2, 20_000), (3, 50_000), (5, 8_000), (8, 120_000)])
def test_bands_reproduce_single_gpu_bitwise(hip, oracle, world, n):
    import torch
    from pedoni_amd.sharded import ShardedModel

    sc = _tall_box(70.0, 210.0)
    field = oracle_field(oracle, sc)
    pos, dest, v0, vel = inject_crowd(field, sc.field.size, n, 2, seed=70 + world)
    vel[:, 1] += np.where(np.arange(n) % 2 == 0, 1.2, -1.2).astype(np.float32)  # cross bands

    def make():
        return hip.HipModel(hip.Options(), sc.field.size, field.distance_map, field.potential_maps,
                            field.unit, sc.obstacle_array())

    single = make()
    single.append(pos, dest, v0, vel)
    single.sort_despawn()

    stream = torch.cuda.current_stream().cuda_stream  # 0 = default stream: one order for all bands
    models = [make() for _ in range(world)]
    cap = 4096
    words = hip.HipModel.halo_bytes(cap) // 4
    sends = [torch.zeros(words, dtype=torch.int32, device="cuda") for _ in range(world)]
    bands = []
    for r, m in enumerate(models):
        m.set_stream(stream)
        bands.append(ShardedModel(m, r, world, halo_cap=cap, gather=lambda s, rv: None,
                                  send=sends[r], recv=sends))
    owner = bands[0].owner_of(pos[:, 1])
    for r, b in enumerate(bands):
        sel = owner == r
        b.load(pos[sel], dest[sel], v0[sel], vel[sel])

    ticks = 12
    for _ in range(ticks):
        single.update_states()
        single.sort_despawn()
    # a band tick is exchange -> sort -> update; one more exchange + sort lines the bands up
    # with `single` (whose loop ends on a sort)
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
    parts = [b.download_owned() for b in bands]
    got = [np.concatenate([p[k] for p in parts]) for k in range(4)]
    assert sum(b.owned_count() for b in bands) == len(want[0]) == len(got[0])
    assert np.array_equal(got[1], want[1])
    for k in (0, 2, 3):
        assert bit_equal(got[k], want[k]).all(), f"array {k} differs between bands and single GPU"
    for m in models + [single]:
        m.close()


def test_split_tick_overlap_form_is_bitwise_identical(hip, oracle):
    """halo_tick_begin / halo_tick_end (edge rows first, pack, interior rows last -- the
    form that overlaps the all-gather with the interior force kernel) against the
    single-model run.  Two send-buffer sets stand in for the asynchronous all-gather."""
    import torch
    from pedoni_amd.sharded import ShardedModel

    world, n = 3, 40_000
    sc = _tall_box(70.0, 210.0)
    field = oracle_field(oracle, sc)
    pos, dest, v0, vel = inject_crowd(field, sc.field.size, n, 2, seed=77)
    vel[:, 1] += np.where(np.arange(n) % 2 == 0, 1.2, -1.2).astype(np.float32)

    def make():
        return hip.HipModel(hip.Options(), sc.field.size, field.distance_map, field.potential_maps,
                            field.unit, sc.obstacle_array())

    single = make()
    single.append(pos, dest, v0, vel)
    single.sort_despawn()
    stream = torch.cuda.current_stream().cuda_stream
    cap = 4096
    words = hip.HipModel.halo_bytes(cap) // 4
    sets = [[torch.zeros(words, dtype=torch.int32, device="cuda") for _ in range(world)] for _ in range(2)]
    models, bands = [make() for _ in range(world)], []
    for r, m in enumerate(models):
        m.set_stream(stream)
        bands.append(ShardedModel(m, r, world, halo_cap=cap, gather=lambda s, rv: None,
                                  send=sets[0][r], recv=sets[0]))
    owner = bands[0].owner_of(pos[:, 1])
    for r, b in enumerate(bands):
        b.load(pos[owner == r], dest[owner == r], v0[owner == r], vel[owner == r])
        b.pack()                                           # first lists -> set 0
    ticks = 10
    for t in range(ticks):
        single.update_states()
        single.sort_despawn()
        cur, nxt = sets[t % 2], sets[(t + 1) % 2]
        for r, b in enumerate(bands):
            below = cur[r - 1].data_ptr() if r > 0 else None
            above = cur[r + 1].data_ptr() if r + 1 < world else None
            b.model.halo_tick_begin(below, above, nxt[r].data_ptr(), cap)
        for b in bands:
            b.model.halo_tick_end()
    cur = sets[ticks % 2]
    for r, b in enumerate(bands):                          # line up with `single` (ends on a sort)
        b.model.halo_unpack(cur[r - 1].data_ptr() if r > 0 else None,
                            cur[r + 1].data_ptr() if r + 1 < world else None, cap)
        b.model.sort_despawn()
    torch.cuda.synchronize()
    want = single.download()
    parts = [b.download_owned() for b in bands]
    got = [np.concatenate([p[k] for p in parts]) for k in range(4)]
    assert sum(b.owned_count() for b in bands) == len(want[0]) == len(got[0])
    assert np.array_equal(got[1], want[1])
    for k in (0, 2, 3):
        assert bit_equal(got[k], want[k]).all()
    for m in models + [single]:
        m.close()


def test_halo_overflow_is_reported(hip, oracle):
    import torch
    from pedoni_amd.sharded import ShardedModel
    from pedoni_amd import abi

    sc = _tall_box(70.0, 120.0)
    field = oracle_field(oracle, sc)
    pos, dest, v0, vel = inject_crowd(field, sc.field.size, 20_000, 2, seed=5)
    stream = torch.cuda.current_stream().cuda_stream  # 0 = default stream: one order for all bands
    models = [hip.HipModel(hip.Options(), sc.field.size, field.distance_map, field.potential_maps,
                           field.unit, sc.obstacle_array()) for _ in range(2)]
    cap = 16      # far too small for a 70 m wide boundary row
    words = hip.HipModel.halo_bytes(cap) // 4
    sends = [torch.zeros(words, dtype=torch.int32, device="cuda") for _ in range(2)]
    bands = []
    for r, m in enumerate(models):
        m.set_stream(stream)
        bands.append(ShardedModel(m, r, 2, halo_cap=cap, gather=lambda s, rv: None,
                                  send=sends[r], recv=sends))
    owner = bands[0].owner_of(pos[:, 1])
    for r, b in enumerate(bands):
        b.load(pos[owner == r], dest[owner == r], v0[owner == r], vel[owner == r])
    for b in bands:
        b.pack()
    for b in bands:
        b.unpack()
        b.finish_tick()
    with pytest.raises(abi.PedoniError, match="overflow"):
        bands[0].owned_count()
    for m in models:
        m.close()
