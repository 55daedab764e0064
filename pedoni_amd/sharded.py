"""Row-band sharding of the pedestrian update over several GPUs.

The reference is single-process; this is the part with no upstream counterpart.  Agents
are kept sorted by neighbor-grid cell in row-major order, so a band of grid rows is a
contiguous index range.  Each rank owns rows [lo, hi) and, every tick,

    pack     its owned agents now in rows {lo-1, lo} ("down") and {hi-1, hi} ("up")
             into one fixed-capacity device buffer          (pedoni_hip_halo_pack)
    exchange one all-gather of those buffers over xGMI       (RCCL; ~0.4 MB per rank)
    unpack   the lower band's UP list in front of its own agents and the upper band's
             DOWN list behind them                          (pedoni_hip_halo_unpack)
    tick     sort/despawn + update_states on rows lo-1 .. hi, integrating lo .. hi-1.

The lists carry both ghosts (agents the neighbour still owns) and migrants (agents that
crossed the band boundary); ownership follows the agent's current row.  Because lower
bands hold lower global indices and the cell sort is stable, every band reproduces the
single-GPU order and therefore the single-GPU result bit for bit.

The exchange is injected (`gather`), so the same driver runs over torch.distributed
(RCCL on GPUs, gloo in the CPU tests) or over an in-process emulation of G bands on one
device.  No data-path computation happens here: only kernel launches through the C-ABI.
"""
from __future__ import annotations

from typing import Callable, List, Optional, Sequence

import numpy as np

from . import abi


def band_rows(n_rows: int, world: int) -> List[int]:
    """Row boundaries of `world` equal bands: band r owns rows [b[r], b[r + 1])."""
    return [(n_rows * r) // world for r in range(world + 1)]


def default_halo_cap(expected_row_agents: int) -> int:
    """A list holds the owned agents of one boundary row plus the few that just crossed it:
    1.5x the expected row population, rounded up to 256 (overflow is detected, not silent)."""
    return max(256, -(-int(1.5 * expected_row_agents) // 256) * 256)


class ShardedModel:
    """One band of a sharded run.

    gather(send, recv): fills recv[r] (r = 0 .. world-1) with rank r's `send`; both are
    objects exposing `.data_ptr()` device addresses (torch tensors, or DeviceBuffer below).
    """

    def __init__(self, model: abi.HipModel, rank: int, world: int, dist=None, torch=None,
                 expected_row_agents: int = 2048, halo_cap: Optional[int] = None,
                 gather: Optional[Callable] = None, send=None, recv: Optional[Sequence] = None,
                 bounds: Optional[Sequence[int]] = None, overlap: bool = False):
        self.model, self.rank, self.world = model, rank, world
        rows, _ = model.neighbor_grid_shape()
        self.bounds = list(bounds) if bounds is not None else band_rows(rows, world)
        self.lo, self.hi = self.bounds[rank], self.bounds[rank + 1]
        if self.hi - self.lo < 2 and world > 1:
            raise abi.PedoniError("a band needs at least two grid rows")
        self.cap = int(halo_cap or default_halo_cap(expected_row_agents))
        self.nbytes = abi.HipModel.halo_bytes(self.cap)
        model.set_band(self.lo, self.hi, self.cap)

        if gather is None:
            if dist is None or torch is None:
                raise abi.PedoniError("ShardedModel needs torch.distributed or a gather callable")
            dev = torch.device("cuda", torch.cuda.current_device()) if torch.cuda.is_available() \
                else torch.device("cpu")
            self._send = torch.zeros(self.nbytes // 4, dtype=torch.int32, device=dev)
            self._recv_flat = torch.zeros(world * (self.nbytes // 4), dtype=torch.int32, device=dev)
            self._recv = [self._recv_flat[r * (self.nbytes // 4):(r + 1) * (self.nbytes // 4)]
                          for r in range(world)]
            if dev.type == "cuda":
                # kernels and the collective are ordered on one stream: no host sync per tick
                model.set_stream(torch.cuda.current_stream().cuda_stream)
            self._async_gather = None
            if dev.type == "cuda" and dist.get_backend() == "nccl" and overlap:
                comm = torch.cuda.Stream()
                main = torch.cuda.current_stream()

                def _async():
                    ev = torch.cuda.Event()
                    ev.record(main)                       # the pack kernel is enqueued
                    comm.wait_event(ev)
                    with torch.cuda.stream(comm):
                        work = dist.all_gather_into_tensor(self._recv_flat, self._send, async_op=True)

                    def _wait():
                        with torch.cuda.stream(main):
                            work.wait()                   # `main` waits on the collective's event
                    return _wait
                self._async_gather = _async
            if dev.type == "cuda" and dist.get_backend() != "nccl":
                # rehearsal backend (gloo): stage through the host; never used for timing
                def _gather(send, recv):
                    host = send.cpu()
                    out = torch.zeros(world * host.numel(), dtype=host.dtype)
                    dist.all_gather_into_tensor(out, host)
                    self._recv_flat.copy_(out)
            else:
                def _gather(send, recv):
                    dist.all_gather_into_tensor(self._recv_flat, send)   # RCCL over xGMI
            self._gather = _gather
        else:
            self._send, self._recv, self._gather = send, list(recv), gather
            self._async_gather = None

    # -- loading -----------------------------------------------------------------------
    def load(self, pos, destination, desired_speed=None, vel=None) -> None:
        """Append this band's own agents (callers split the crowd by `owner_of`) and
        establish the sorted order the first pack needs."""
        if len(pos):
            self.model.append(pos, destination, desired_speed, vel)
        self.model.sort_despawn()

    def owner_of(self, pos_y: np.ndarray, grid_unit: float = 1.4) -> np.ndarray:
        # (pos / unit).as_ivec2(): truncation toward zero (neighbor_grid.rs:27)
        rows = np.trunc(np.asarray(pos_y, np.float32) / np.float32(grid_unit)).astype(np.int64)
        return np.searchsorted(np.asarray(self.bounds[1:-1]), rows, side="right")

    # -- stepping ------------------------------------------------------------------------
    def pack(self) -> None:
        self.model.halo_pack(self._send.data_ptr(), self.cap)

    def unpack(self) -> None:
        below = self._recv[self.rank - 1].data_ptr() if self.rank > 0 else None
        above = self._recv[self.rank + 1].data_ptr() if self.rank + 1 < self.world else None
        self.model.halo_unpack(below, above, self.cap)

    def exchange(self) -> None:
        self.pack()
        self._gather(self._send, self._recv)
        self.unpack()

    def finish_tick(self) -> None:
        self.model.sort_despawn()
        self.model.update_states()

    def tick(self) -> None:
        self.exchange()
        self.finish_tick()

    def tick_n(self, steps: int) -> None:
        """`steps` ticks.  Per tick: the all-gather of the lists packed by the previous
        tick, then ONE fused call (unpack, sort/despawn, update_states, pack).  With
        `overlap` (opt-in, RCCL runs) the fused call is split: the rows beside the band's
        edges are updated and packed first, the all-gather for the NEXT tick is started on
        a side stream, and the interior rows -- the bulk of the work -- run meanwhile.
        Measured on one MI355X with a 1-rank RCCL group (no wire latency): plain form
        +15 us per tick over the unsharded tick, overlap form +47 us (cross-stream event
        waits, two kernels sharing the CUs) -- so overlap only pays once the all-gather
        itself costs more than ~35 us; the default is the plain form."""
        if steps <= 0:
            return
        m = self.model
        below = self._recv[self.rank - 1].data_ptr() if self.rank > 0 else None
        above = self._recv[self.rank + 1].data_ptr() if self.rank + 1 < self.world else None
        self.pack()
        if self._async_gather is None:
            for _ in range(steps):
                self._gather(self._send, self._recv)
                m.halo_tick(below, above, self._send.data_ptr(), self.cap)
            return
        pending = self._async_gather()
        for _ in range(steps):
            pending()                                   # this stream waits for the lists
            m.halo_tick_begin(below, above, self._send.data_ptr(), self.cap)
            pending = self._async_gather()              # next tick's lists fly ...
            m.halo_tick_end()                           # ... while the interior is computed
        pending()

    def owned_count(self) -> int:
        return self.model.owned_count()

    def download_owned(self):
        """Owned agents (rows lo .. hi-1) of the current sorted order, ghosts stripped."""
        pos, dest, vel, v0 = self.model.download()
        with np.errstate(invalid="ignore"):
            rows = np.trunc(np.nan_to_num(pos[:, 1] / np.float32(self.model.options.neighbor_grid_unit),
                                          nan=-1e9)).astype(np.int64)
            keep = (rows >= self.lo) & (rows < self.hi) & ~np.isnan(pos).any(axis=1)
        return pos[keep], dest[keep], vel[keep], v0[keep]
