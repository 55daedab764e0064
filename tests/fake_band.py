"""Test double for one band of a sharded run, backed by the CPU oracle (tests only).

Implements the slice of the HipModel interface that pedoni_amd.sharded.ShardedModel drives
(set_band / append / sort_despawn / update_states / halo_pack / halo_unpack / halo_tick /
owned_count / download) with the SAME buffer layout as include/pedoni_hip.h, so the
world-size-2 gloo test exercises the real orchestration, band split and exchange protocol
on CPU.  The GPU kernels themselves are covered by tests/test_gpu_sharded.py.
"""
import ctypes
import socket
from types import SimpleNamespace

import numpy as np

HEADER, RECORD = 4, 6


def _view(ptr, n_words):
    return np.ctypeslib.as_array((ctypes.c_uint32 * n_words).from_address(ptr))


class OracleBandModel:
    def __init__(self, oracle, size, field, grid_unit=1.4):
        self.oracle, self.size, self.field, self.unit = oracle, size, field, np.float32(grid_unit)
        self.options = SimpleNamespace(neighbor_grid_unit=grid_unit)
        self.rows, self.cols = oracle.neighbor_grid_shape(size, grid_unit)
        self.lo, self.hi, self.cap = 0, self.rows, 0
        self.own = self._empty()
        self.below = self.above = None
        self.m = None

    @staticmethod
    def _empty():
        return (np.zeros((0, 2), np.float32), np.zeros(0, np.uint32), np.zeros((0, 2), np.float32),
                np.zeros(0, np.float32))

    def _row(self, pos):
        with np.errstate(invalid="ignore"):
            return np.trunc(np.nan_to_num(pos[:, 1] / self.unit, nan=-1e9)).astype(np.int64)

    # -- HipModel surface ----------------------------------------------------------------
    def neighbor_grid_shape(self):
        return self.rows, self.cols

    def set_band(self, lo, hi, cap):
        self.lo, self.hi, self.cap = lo, hi, cap

    def append(self, pos, dest, v0, vel):
        new = (np.asarray(pos, np.float32).reshape(-1, 2), np.asarray(dest, np.uint32),
               np.asarray(vel, np.float32).reshape(-1, 2), np.asarray(v0, np.float32))
        self.own = tuple(np.concatenate([a, b]) for a, b in zip(self.own, new))

    def sort_despawn(self):
        parts = [p for p in (self.below, self.own, self.above) if p is not None]
        pos, dest, vel, v0 = (np.concatenate([p[k] for p in parts]) for k in range(4))
        r = self._row(pos)
        keep = (r >= self.lo - 1) & (r <= self.hi)
        self.m = self.oracle.OracleModel(self.size, neighbor_grid_unit=float(self.unit))
        self.m.spawn_pedestrians(self.field, pos[keep], dest[keep], v0[keep], vel[keep])
        self.below = self.above = None
        self.sorted_state = self.m.download()

    def update_states(self):
        r = self._row(self.sorted_state[0])
        owned = (r >= self.lo) & (r < self.hi)
        self.m.update_states(self.field)
        pos, dest, vel, v0 = self.m.download()
        self.own = (pos[owned], dest[owned], vel[owned], v0[owned])

    def _write_list(self, buf, sel):
        pos, dest, vel, v0 = self.own
        n = int(sel.sum())
        assert n <= self.cap, "halo list overflow in the test double"
        buf[0], buf[1], buf[2], buf[3] = n, 0, 0, 0
        rec = np.zeros((n, RECORD), np.uint32)
        rec[:, 0:2] = pos[sel].view(np.uint32)
        rec[:, 2:4] = vel[sel].view(np.uint32)
        rec[:, 4] = v0[sel].view(np.uint32)
        rec[:, 5] = dest[sel]
        buf[HEADER:HEADER + n * RECORD] = rec.ravel()

    def halo_pack(self, send_ptr, cap):
        words = HEADER + cap * RECORD
        buf = _view(send_ptr, 2 * words)
        r = self._row(self.own[0])
        none = np.zeros(len(r), bool)
        self._write_list(buf[:words], ((r == self.lo - 1) | (r == self.lo)) if self.lo > 0 else none)
        self._write_list(buf[words:], ((r == self.hi - 1) | (r == self.hi)) if self.hi < self.rows else none)

    def _read_list(self, buf):
        n = int(buf[0])
        rec = np.array(buf[HEADER:HEADER + n * RECORD]).reshape(n, RECORD)
        return (rec[:, 0:2].copy().view(np.float32), rec[:, 5].copy(),
                rec[:, 2:4].copy().view(np.float32), rec[:, 4].copy().view(np.float32))

    def halo_unpack(self, below_ptr, above_ptr, cap):
        words = HEADER + cap * RECORD
        self.below = self._read_list(_view(below_ptr, 2 * words)[words:]) if below_ptr else None
        self.above = self._read_list(_view(above_ptr, 2 * words)[:words]) if above_ptr else None

    def halo_tick(self, below_ptr, above_ptr, send_ptr, cap):
        self.halo_unpack(below_ptr, above_ptr, cap)
        self.sort_despawn()
        self.update_states()
        self.halo_pack(send_ptr, cap)

    def owned_count(self):
        return len(self.own[0])

    def download(self):
        return self.m.download()


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def band_worker(rank, world, port, ticks, out_dir):
    """torch.multiprocessing entry: one band per process over gloo on 127.0.0.1."""
    import os
    import sys
    from pathlib import Path
    root = Path(__file__).resolve().parent.parent
    for p in (str(root), str(root / "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import torch
    import torch.distributed as dist
    from oracle import pyoracle as oracle
    import helpers
    from pedoni_amd.sharded import ShardedModel

    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        sc, pos, dest, v0, vel = sharded_case()
        field = helpers.oracle_field(oracle, sc)
        model = OracleBandModel(oracle, sc.field.size, field)
        band = ShardedModel(model, rank, world, dist, torch, halo_cap=2048)
        mine = band.owner_of(pos[:, 1]) == rank
        band.load(pos[mine], dest[mine], v0[mine], vel[mine])
        band.tick_n(ticks)
        np.savez(Path(out_dir) / f"band{rank}.npz", pos=model.own[0], dest=model.own[1],
                 vel=model.own[2], v0=model.own[3], bounds=np.array(band.bounds))
    finally:
        dist.destroy_process_group()


def sharded_case():
    import helpers
    from pedoni_amd import scenario as scn
    sc = scn.Scenario()
    sc.field = scn.FieldConfig((40.0, 90.0))
    sc.waypoints = [scn.SegmentConfig(((4, 4), (4, 86))), scn.SegmentConfig(((36, 4), (36, 86)))]
    sc.obstacles = [scn.SegmentConfig(((0, 0), (0, 90)), 0.2), scn.SegmentConfig(((40, 0), (40, 90)), 0.2),
                    scn.SegmentConfig(((0, 0), (40, 0)), 0.2), scn.SegmentConfig(((0, 90), (40, 90)), 0.2)]
    from oracle import pyoracle as oracle
    field = helpers.oracle_field(oracle, sc)
    pos, dest, v0, vel = helpers.inject_crowd(field, sc.field.size, 4000, 2, seed=91)
    vel[:, 1] += np.where(np.arange(len(pos)) % 2 == 0, 1.0, -1.0).astype(np.float32)
    return sc, pos, dest, v0, vel
