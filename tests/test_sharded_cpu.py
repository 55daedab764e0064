"""N > 1 path on CPU: two processes over gloo drive pedoni_amd.sharded.ShardedModel (band
split, fixed-capacity halo buffers, all-gather, ghost + migrant protocol) with an
oracle-backed band model; the union of the bands must equal the unsharded oracle run."""
import numpy as np
import pytest

import fake_band
from helpers import bit_equal, oracle_field
from pedoni_amd.sharded import band_rows, default_halo_cap


def test_band_rows_cover_the_grid():
    for rows in (16, 150, 5715):
        for world in (1, 2, 3, 8):
            b = band_rows(rows, world)
            assert b[0] == 0 and b[-1] == rows and all(x < y for x, y in zip(b, b[1:]))
    assert default_halo_cap(1400) % 256 == 0 and default_halo_cap(1400) >= 2100


@pytest.mark.timeout(300)
def test_two_bands_over_gloo_equal_unsharded_run(oracle, tmp_path):
    import torch.multiprocessing as mp
    ticks, world = 6, 2
    mp.spawn(fake_band.band_worker, args=(world, fake_band.free_port(), ticks, str(tmp_path)),
             nprocs=world, join=True)

    sc, pos, dest, v0, vel = fake_band.sharded_case()
    field = oracle_field(oracle, sc)
    ref = oracle.OracleModel(sc.field.size)
    ref.spawn_pedestrians(field, pos, dest, v0, vel)
    for t in range(ticks):
        if t:
            ref.spawn_pedestrians(field)
        ref.update_states(field)
    wp, wd, wv, w0 = ref.download()

    parts = [np.load(tmp_path / f"band{r}.npz") for r in range(world)]
    gp = np.concatenate([p["pos"] for p in parts]); gv = np.concatenate([p["vel"] for p in parts])
    g0 = np.concatenate([p["v0"] for p in parts]); gd = np.concatenate([p["dest"] for p in parts])
    assert len(gp) == len(wp), "agents lost or duplicated across the band boundary"
    # some agent crossed the boundary in each direction, or the test proves nothing
    bounds = parts[0]["bounds"]
    start0 = set(v0[np.trunc(pos[:, 1] / np.float32(1.4)) < bounds[1]].view(np.uint32).tolist())
    end0 = set(parts[0]["v0"].view(np.uint32).tolist())
    assert end0 - start0 and start0 - end0, "no agent migrated between the bands"
    order_g, order_w = np.argsort(g0.view(np.uint32), kind="stable"), np.argsort(w0.view(np.uint32), kind="stable")
    assert np.array_equal(g0[order_g].view(np.uint32), w0[order_w].view(np.uint32))
    assert np.array_equal(gd[order_g], wd[order_w])
    assert bit_equal(gp[order_g], wp[order_w]).all()
    assert bit_equal(gv[order_g], wv[order_w]).all()


# ---- the periodic re-cut of the bands: pure host code (pedoni_shard_recut_bounds) -------------
from hypothesis import given, settings, strategies as st   # noqa: E402


def _loads(bounds, counts):
    return [int(counts[bounds[i]:bounds[i + 1]].sum()) for i in range(len(bounds) - 1)]


@settings(max_examples=150, deadline=None)
@given(st.integers(2, 8), st.integers(60, 400), st.integers(1, 6), st.integers(0, 2 ** 31), st.integers(-1, 20))
def test_recut_keeps_every_invariant(world, n_rows, max_shift, seed, slack):
    """Whatever the crowd: boundaries stay ordered with >= 6 rows per band, move <= max_shift
    rows, hand over no more agents than one bulk list holds, never leave the slack the map
    slices were cut with, and outer boundaries never move."""
    from pedoni_amd import abi
    rng = np.random.default_rng(seed)
    counts = (rng.gamma(0.6, 200.0, n_rows) * (rng.random(n_rows) < 0.8)).astype(np.uint32)
    bounds0 = [(n_rows * r) // world for r in range(world + 1)]
    bulk_cap = int(rng.integers(100, 5000))
    bounds = list(bounds0)
    for _ in range(12):
        nb = abi.recut_bounds(bounds, counts, max_shift, bulk_cap, bounds0, slack)
        assert nb[0] == 0 and nb[-1] == n_rows
        assert all(nb[i + 1] - nb[i] >= 6 for i in range(world))
        for b in range(1, world):
            old, to = bounds[b], nb[b]
            assert abs(to - old) <= max_shift
            if slack >= 0:
                assert abs(to - bounds0[b]) <= slack
            if to < old:      # donor below sends rows [to-1, old-1)
                assert counts[max(to - 1, 0):old - 1].sum() <= bulk_cap
                assert to - 1 >= bounds[b - 1]
            elif to > old:    # donor above sends rows [old+1, to+1)
                assert counts[old + 1:to + 1].sum() <= bulk_cap
                assert to + 1 <= bounds[b + 1]
        bounds = nb


def test_recut_converges_towards_equal_loads():
    from pedoni_amd import abi
    n_rows, world = 600, 4
    counts = np.full(n_rows, 10, np.uint32)
    counts[:150] = 300                                    # three quarters of the crowd in the first quarter
    bounds = [(n_rows * r) // world for r in range(world + 1)]
    first = max(_loads(bounds, counts)) / (counts.sum() / world)
    for _ in range(200):
        bounds = abi.recut_bounds(bounds, counts, 4, 100_000)
    last = max(_loads(bounds, counts)) / (counts.sum() / world)
    assert first > 3.0 and last < 1.2, (first, last, bounds)
    assert bounds == abi.recut_bounds(bounds, counts, 4, 100_000)      # a fixed point
