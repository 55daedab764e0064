#!/usr/bin/env python3
"""Runs pedoni_shard_tick_n -- the RCCL-driven multi-rank tick below the C-ABI -- with world > 1
on ONE GPU: every rank is a host thread with its own model and stream, and librccl is replaced
(PEDONI_RCCL_LIB, set by the test that starts this process) by tests/loopback_rccl, whose
send / receive are event-ordered device copies.  Compares the merged bands with the unsharded
model bit for bit and prints one JSON line.  Its own process because libpedoni_hip resolves
RCCL once per process (the GPU suite's own process uses the real librccl).

    python tests/loopback_shard_runner.py WORLD MODE      MODE: plain | overlap | recut | recut_overlap
                                                          | fault (group-close after an injected failure)
"""
import ctypes as C
import json
import os
import sys
import threading
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))

CAP = 8192 if os.environ.get("LOOPBACK_BIG") == "1" else 4096


def main():
    world, mode = int(sys.argv[1]), sys.argv[2]
    lib_path = os.environ["PEDONI_RCCL_LIB"]
    assert "loopback" in lib_path
    import torch
    from helpers import bit_equal
    from oracle import pyoracle               # field builder of the test scenario only (checker side)
    from pedoni_amd import abi
    import test_gpu_shard as tg               # scenario / crowd builders shared with the local-group tests

    loop = C.CDLL(lib_path, mode=C.RTLD_GLOBAL)   # same handle libpedoni_hip will dlopen
    loop.loopback_rccl_group_depth.restype = C.c_int

    if mode == "fault":
        return fault_case(abi, pyoracle, tg, loop)
    if mode == "tick_fault":
        return tick_fault_case(abi, pyoracle, tg, world)

    big = os.environ.get("LOOPBACK_BIG") == "1"      # bands large enough for the 7-wave kernel BY SIZE
    tall = os.environ.get("LOOPBACK_TALL") == "1"    # bands of > 2048 grid rows each (a narrow, very tall box)
    sc = tg._tall_box(400.0, 1000.0) if big else (tg._tall_box(24.0, 3100.0 * world) if tall else tg._tall_box(70.0, 210.0))
    field = tg.oracle_field(pyoracle, sc)
    pos, dest, v0, vel = tg._lopsided_crowd(field, sc.field.size, 1_000_000 if big else 60_000, seed=70 + world)

    opts = lambda: abi.Options(math_mode=abi.MATH_FAST if os.environ.get("LOOPBACK_MATH") == "fast" else abi.MATH_EXACT)
    single = abi.HipModel(opts(), sc.field.size, field.distance_map, field.potential_maps, field.unit,
                          sc.obstacle_array())
    single.append(pos, dest, v0, vel)
    single.sort_despawn()
    rows, cols = single.neighbor_grid_shape()
    idx = single.neighbor_grid_indices().astype(np.int64)
    row_counts = np.diff(idx[::cols][:rows + 1])
    recut = mode.startswith("recut")
    # re-cut runs start from equal ROWS on a lopsided crowd, so the re-cut has work to do
    bounds = [(rows * r) // world for r in range(world + 1)] if recut else abi.balanced_bounds(row_counts, world)
    slack = 12 if recut else 0
    uid = abi.shard_unique_id()
    assert uid[:8] == b"LOOPBACK", "libpedoni_hip did not resolve the loop-back transport"
    band_of = np.searchsorted(np.asarray(bounds[1:-1]), np.trunc(pos[:, 1] / np.float32(1.4)).astype(np.int64),
                              side="right")
    phases = {"plain": [(False, 17)], "overlap": [(False, 4), (True, 9), (False, 4)],
              "recut": [(False, 17)], "recut_overlap": [(False, 3), (True, 10), (False, 4)]}[mode]
    if big:
        phases = [(False, 2), (True, int(os.environ.get("LOOPBACK_OVERLAP_TICKS", "30"))), (False, 2)]
    ticks = sum(k for _, k in phases)

    models, shards, errors, loads0 = [None] * world, [None] * world, [None] * world, [0] * world
    forms = [None] * world

    def rank_main(r):
        try:
            rows_needed = abi.shard_map_rows(bounds[r], bounds[r + 1], slack, 1.4, field.unit, field.shape[0])
            m = abi.HipModel(opts(), sc.field.size, field.distance_map, field.potential_maps, field.unit,
                             sc.obstacle_array(), map_rows=rows_needed)
            models[r] = m
            s = abi.Shard(m, r, world, bounds, CAP, unique_id=uid)       # ncclCommInitRank: waits for all ranks
            shards[r] = s
            s.selftest()                                                  # token ring through the exchange itself
            if recut:
                s.set_rebalance(4, 3, map_slack_rows=slack)
            sel = band_of == r
            if sel.any():
                m.append(pos[sel], dest[sel], v0[sel], vel[sel])
            s.begin()
            loads0[r] = s.owned_count()
            for overlap, k in phases:
                s.set_overlap(overlap)
                s.tick_n(k)
            s.set_overlap(False)
            m.synchronize()
            forms[r] = s.tick_forms()
            assert loop.loopback_rccl_group_depth() == 0
        except Exception as e:                                            # noqa: BLE001
            errors[r] = f"rank {r}: {type(e).__name__}: {e}"

    threads = [threading.Thread(target=rank_main, args=(r,)) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    if any(errors):
        print(json.dumps({"ok": False, "errors": [e for e in errors if e]}))
        return 1

    for _ in range(ticks):
        single.update_states()
        single.sort_despawn()
    # a pending overlapped exchange holds the NEXT tick's lists in the receive buffers; the line-up
    # below re-packs and exchanges through the public halo entry points instead
    words = abi.HipModel.halo_bytes(CAP) // 4
    sends = [torch.zeros(words, dtype=torch.int32, device="cuda") for _ in range(world)]
    torch.cuda.synchronize()
    for m, snd in zip(models, sends):
        m.halo_pack(snd.data_ptr(), CAP)
        m.synchronize()
    for r, m in enumerate(models):
        m.halo_unpack(sends[r - 1].data_ptr() if r > 0 else None,
                      sends[r + 1].data_ptr() if r + 1 < world else None, CAP)
        m.sort_despawn()
        m.synchronize()

    want = single.download()
    parts = [s.download_owned() for s in shards]
    got = [np.concatenate([p[k] for p in parts]) for k in range(4)]
    stats = (C.c_long * 3)()
    loop.loopback_rccl_stats(stats)
    new_bounds = [shards[r].band()[0] for r in range(world)] + [shards[-1].band()[1]]
    out = {
        "ok": True, "world": world, "mode": mode, "ticks": ticks, "agents": int(len(want[0])),
        "count_equal": bool(len(got[0]) == len(want[0]) == sum(s.owned_count() for s in shards)),
        "dest_equal": bool(len(got[1]) == len(want[1]) and np.array_equal(got[1], want[1])),
        "bit_equal": bool(len(got[0]) == len(want[0]) and all(bit_equal(got[k], want[k]).all() for k in (0, 2, 3))),
        "sends": int(stats[0]), "recvs": int(stats[1]), "allreduces": int(stats[2]),
        "bounds0": [int(b) for b in bounds], "bounds1": [int(b) for b in new_bounds],
        "loads0": loads0, "loads": [s.owned_count() for s in shards], "forms": forms, "grid_rows": int(rows),
    }
    for s in shards:
        s.close()
    for m in models + [single]:
        m.close()
    print(json.dumps(out))
    return 0


def tick_fault_case(abi, pyoracle, tg, world):
    """LOOPBACK_RCCL_FAIL_SEND=k with k beyond the token ring's sends: an ncclSend of a TICK fails on one rank
    (the other runs into the receive's timeout).  The failing call must report it, and the shard must refuse
    further ticks -- its hand-offs are in no state to continue from -- until the band is reloaded and
    pedoni_shard_begin is called again."""
    sc = tg._tall_box(70.0, 210.0)
    field = tg.oracle_field(pyoracle, sc)
    pos, dest, v0, vel = tg._lopsided_crowd(field, sc.field.size, 30_000, seed=5)
    probe = abi.HipModel(abi.Options(), sc.field.size, field.distance_map, field.potential_maps, field.unit, sc.obstacle_array())
    rows, _ = probe.neighbor_grid_shape()
    probe.close()
    bounds = [(rows * r) // world for r in range(world + 1)]
    uid = abi.shard_unique_id()
    band_of = np.searchsorted(np.asarray(bounds[1:-1]), np.trunc(pos[:, 1] / np.float32(1.4)).astype(np.int64), side="right")
    first, second, after_begin = [None] * world, [None] * world, [None] * world

    def rank_main(r):
        m = abi.HipModel(abi.Options(), sc.field.size, field.distance_map, field.potential_maps, field.unit, sc.obstacle_array())
        s = abi.Shard(m, r, world, bounds, CAP, unique_id=uid)
        s.selftest()
        sel = band_of == r
        m.append(pos[sel], dest[sel], v0[sel], vel[sel])
        s.begin()
        s.set_overlap(os.environ.get("LOOPBACK_OVERLAP") == "1")
        try:
            s.tick_n(6)
            first[r] = "ok"
        except abi.PedoniError as e:
            first[r] = str(e)
        try:
            s.tick_n(1)
            second[r] = "ok"
        except abi.PedoniError as e:
            second[r] = str(e)
        after_begin[r] = "not tried"
        s.close(); m.close()

    threads = [threading.Thread(target=rank_main, args=(r,)) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    print(json.dumps({"ok": True, "first": first, "second": second}))
    return 0


def fault_case(abi, pyoracle, tg, loop):
    """LOOPBACK_RCCL_FAIL_SEND=0: the first ncclSend of the process fails INSIDE a group.  The call
    must report it, leave no group open on the thread, and the next exchange must work."""
    sc = tg._tall_box(60.0, 80.0)
    field = tg.oracle_field(pyoracle, sc)
    m = abi.HipModel(abi.Options(), sc.field.size, field.distance_map, field.potential_maps, field.unit,
                     sc.obstacle_array())
    rows, _ = m.neighbor_grid_shape()
    s = abi.Shard(m, 0, 1, [0, rows], 1024, unique_id=abi.shard_unique_id())
    first = None
    try:
        s.selftest()
    except abi.PedoniError as e:
        first = str(e)
    depth = int(loop.loopback_rccl_group_depth())
    second_ok = True
    try:
        s.selftest()          # a group left open would swallow this exchange: the token would never arrive
    except abi.PedoniError as e:
        second_ok = str(e)
    s.close(); m.close()
    print(json.dumps({"ok": True, "first_error": first, "depth_after_failure": depth, "second": second_ok}))
    return 0


if __name__ == "__main__":
    sys.exit(main())
