"""Heaviest tiles first (kernels.hpp build_tile_order): the sort pass turns the weights every wave of the last
force launch left into the next launch's workgroup order.  Placement only -- so: it is a permutation that
keeps every XCD on its own contiguous eighth of the agents, heavier classes come first, without weights the
order is the plain one, and not a bit of the results moves (with / without it, and against the oracle)."""
import numpy as np
import pytest

from helpers import bit_equal, box_scenario, inject_crowd, oracle_field

pytestmark = pytest.mark.gpu


def _make(hip, sc, field):
    return hip.HipModel(hip.Options(), sc.field.size, field.distance_map, field.potential_maps, field.unit, sc.obstacle_array())


def _chunks(n_blocks):
    q, r = divmod(n_blocks, 8)
    first = [x * (q + 1) if x < r else r * (q + 1) + (x - r) * q for x in range(8)]
    return first, [q + (1 if x < r else 0) for x in range(8)]


def _lopsided(field, size, n, seed):
    """A crowd with a dense blob (4x the density) in one corner region: its tiles are the heavy ones."""
    pos, dest, v0, vel = inject_crowd(field, size, n, 2, seed=seed, clearance=0.6, min_potential=1.0)
    rng = np.random.default_rng(seed)
    k = n // 4
    pos[:k] = (np.array([0.55, 0.6]) * size + rng.uniform(-0.06, 0.06, (k, 2)) * size).astype(np.float32)
    return pos, dest, v0, vel


def test_order_is_a_chunk_preserving_permutation_with_the_heavy_tiles_first(hip, oracle, monkeypatch):
    monkeypatch.setenv("PEDONI_FORCE_GROUP", "1")          # one lane per agent: the launch the order is for
    sc = box_scenario(240.0)
    field = oracle_field(oracle, sc)
    pos, dest, v0, vel = _lopsided(field, sc.field.size, 120_000, seed=3)
    m = _make(hip, sc, field)
    m.append(pos, dest, v0, vel)
    m.tick_n(1)                  # pass 1 sees no weights yet: the plain order
    order, _ = m.tile_order()
    n_blocks = len(order)
    assert n_blocks == (m.get_pedestrian_count() + 255) // 256 or n_blocks >= 460
    first, length = _chunks(n_blocks)
    plain = np.empty(n_blocks, np.uint32)
    for x in range(8):
        plain[x + 8 * np.arange(length[x])] = first[x] + np.arange(length[x])
    assert np.array_equal(order, plain), "without weights the order must be the XCD-contiguous one"
    m.tick_n(3)
    m.sort_despawn()             # builds the order from the weights of the last launch
    order, weight = m.tile_order()
    assert np.array_equal(np.sort(order), np.arange(n_blocks)), "not a permutation"
    heavy_seen = 0
    for x in range(8):
        tiles = order[x + 8 * np.arange(length[x])].astype(np.int64)
        assert ((tiles >= first[x]) & (tiles < first[x] + length[x])).all(), "an XCD left its contiguous chunk"
        # the builder's classes, restated: a tile ranks by the heaviest of itself and its two neighbours in the
        # chunk, in classes a quarter of the chunk's mean weight wide, centred on the mean
        w = weight[first[x]:first[x] + length[x]].astype(np.int64)
        rank = np.maximum(w, np.maximum(np.r_[0, w[:-1]], np.r_[w[1:], 0]))
        total = max(int(w.sum()), 1)
        level = np.minimum((8 * rank * length[x] + total) // (2 * total), 63)
        cls = 63 - level
        c = cls[tiles - first[x]]
        assert (np.diff(c) >= 0).all(), "a lighter class before a heavier one"
        heavy_seen += int((level >= 7).sum())
        if (level >= 7).any():
            assert level[tiles[0] - first[x]] >= 7        # an XCD that has heavy tiles starts with one
    assert heavy_seen > 20, "the blob's tiles should stand out"
    m.close()


def test_results_do_not_depend_on_the_order(hip, oracle, monkeypatch):
    monkeypatch.setenv("PEDONI_FORCE_GROUP", "1")
    sc = box_scenario(240.0)
    field = oracle_field(oracle, sc)
    pos, dest, v0, vel = _lopsided(field, sc.field.size, 100_000, seed=8)
    with_order = _make(hip, sc, field)
    monkeypatch.setenv("PEDONI_NO_TILE_ORDER", "1")
    plain = _make(hip, sc, field)
    monkeypatch.delenv("PEDONI_NO_TILE_ORDER")
    cpu = oracle.OracleModel(sc.field.size)
    cpu.spawn_pedestrians(field, pos, dest, v0, vel)
    for m in (with_order, plain):
        m.append(pos, dest, v0, vel)
        m.spawn_pedestrians()
    for tick in range(5):
        for m in (with_order, plain):
            m.update_states(); m.spawn_pedestrians()
        cpu.update_states(field); cpu.spawn_pedestrians(field)
    assert len(with_order.tile_order()[0]) and not len(plain.tile_order()[0])
    order = with_order.tile_order()[0]
    first, length = _chunks(len(order))
    assert any(not np.array_equal(order[x + 8 * np.arange(length[x])], first[x] + np.arange(length[x])) for x in range(8)), \
        "the lopsided crowd should have re-ordered some XCD's tiles"
    for x, y, z in zip(with_order.download(), plain.download(), cpu.download()):
        same = (lambda a, b: a.shape == b.shape and (bit_equal(a, b).all() if a.dtype == np.float32 else np.array_equal(a, b)))
        assert same(x, y) and same(x, z)
    # the replayed graph carries the order too: 6 more ticks without a host sync in between
    with_order.tick_n(6); plain.tick_n(6)
    for x, y in zip(with_order.download(), plain.download()):
        assert x.shape == y.shape and (bit_equal(x, y).all() if x.dtype == np.float32 else np.array_equal(x, y))
    with_order.close(); plain.close()
