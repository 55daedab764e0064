"""ctypes binding of include/pedoni_hip.h (the C-ABI of the gfx950 backend).

Thin by design: every method forwards to one ``pedoni_hip_*`` entry point and raises
``PedoniError`` on a non-zero status.  No computation happens in Python and nothing
falls back to a CPU path: if ``libpedoni_hip.so`` is missing or no GPU is present the
calls fail loudly.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass
from pathlib import Path
from typing import Optional, Sequence

import numpy as np

_ROOT = Path(__file__).resolve().parent
_LIB: Optional[C.CDLL] = None

N_KERNELS = 8
K_BIN, K_SCAN, K_SLOT, K_REORDER, K_FORCE, K_HALO_PACK, K_HALO_UNPACK, K_OTHER = range(8)
MATH_EXACT = 0
MATH_FAST = 1
HALO_HEADER_WORDS = 4
HALO_RECORD_WORDS = 6

# every symbol include/pedoni_hip.h declares (tests check the library exports them all)
SYMBOLS = [
    "pedoni_hip_last_error", "pedoni_hip_default_options", "pedoni_hip_device_count",
    "pedoni_hip_create", "pedoni_hip_destroy", "pedoni_hip_spawn_pedestrians",
    "pedoni_hip_update_states", "pedoni_hip_list_pedestrians",
    "pedoni_hip_get_pedestrian_count", "pedoni_hip_append", "pedoni_hip_sort_despawn",
    "pedoni_hip_tick_n", "pedoni_hip_tick", "pedoni_hip_download", "pedoni_hip_clear",
    "pedoni_hip_neighbor_grid_indices", "pedoni_hip_neighbor_grid_shape", "pedoni_hip_cell_flags", "pedoni_hip_tile_order",
    "pedoni_hip_calc_accelerations", "pedoni_hip_set_stream", "pedoni_hip_get_stream",
    "pedoni_hip_synchronize", "pedoni_hip_profile", "pedoni_hip_kernel_times",
    "pedoni_hip_kernel_name", "pedoni_hip_set_band", "pedoni_hip_halo_bytes",
    "pedoni_hip_halo_pack", "pedoni_hip_halo_unpack", "pedoni_hip_halo_tick",
    "pedoni_hip_halo_tick_begin", "pedoni_hip_halo_tick_end",
    "pedoni_hip_owned_count",
    "pedoni_hip_selftest_math", "pedoni_hip_selftest_pair", "pedoni_hip_selftest_field", "pedoni_hip_set_spawners", "pedoni_hip_get_spawn_rng", "pedoni_hip_set_speed_rng",
    "pedoni_hip_profile_every", "pedoni_hip_profile_burst", "pedoni_hip_force_kernel_info",
    "pedoni_hip_create_rows", "pedoni_shard_map_rows", "pedoni_hip_eikonal",
    "pedoni_shard_unique_id", "pedoni_shard_balanced_bounds", "pedoni_shard_recut_bounds", "pedoni_shard_create", "pedoni_shard_destroy",
    "pedoni_shard_begin", "pedoni_shard_tick_n", "pedoni_shard_owned_count", "pedoni_shard_band",
    "pedoni_shard_selftest", "pedoni_shard_set_rebalance", "pedoni_shard_set_overlap", "pedoni_shard_tick_forms", "pedoni_shard_local_group_tick_n",
]


# the diagnostics build (pedoni_amd/lib/libpedoni_hip_diag.so, -DPEDONI_DIAGNOSTICS) adds these
DIAG_SYMBOLS = ["pedoni_hip_debug_set_status", "pedoni_hip_debug_set_ablate", "pedoni_hip_debug_force_trace",
                "pedoni_hip_debug_force_trace_raw"]


class PedoniError(RuntimeError):
    pass


class _Options(C.Structure):
    _fields_ = [
        ("neighbor_grid_unit", C.c_float), ("field_grid_unit", C.c_float),
        ("use_neighbor_grid", C.c_int32), ("use_distance_map", C.c_int32),
        ("gpu_work_size", C.c_int32), ("math_mode", C.c_int32),
        ("seed", C.c_uint64), ("initial_capacity", C.c_uint32), ("reserved", C.c_uint32),
    ]


class _Obstacle(C.Structure):
    _fields_ = [("x0", C.c_float), ("y0", C.c_float), ("x1", C.c_float), ("y1", C.c_float),
                ("width", C.c_float)]


class _Pedestrian(C.Structure):
    _fields_ = [("x", C.c_float), ("y", C.c_float), ("destination", C.c_uint64)]


class _StepMetrics(C.Structure):
    _fields_ = [("active_ped_count", C.c_int32), ("time_spawn", C.c_double),
                ("time_calc_state", C.c_double), ("time_calc_state_kernel", C.c_double)]


class _KernelTimes(C.Structure):
    _fields_ = [("total_ms", C.c_double * N_KERNELS), ("launches", C.c_uint64 * N_KERNELS)]


PED_DTYPE = np.dtype([("x", "<f4"), ("y", "<f4"), ("destination", "<u8")], align=True)


@dataclass
class Options:
    """lib.rs:108-135 `SimulatorOptions` (same defaults) + backend knobs."""
    neighbor_grid_unit: float = 1.4
    field_grid_unit: float = 0.25
    use_neighbor_grid: bool = True
    use_distance_map: bool = True
    gpu_work_size: int = 0
    math_mode: int = MATH_EXACT
    seed: int = 12345
    initial_capacity: int = 0

    def _c(self) -> _Options:
        return _Options(self.neighbor_grid_unit, self.field_grid_unit,
                        int(self.use_neighbor_grid), int(self.use_distance_map),
                        self.gpu_work_size, self.math_mode, self.seed, self.initial_capacity, 0)


def library_path() -> Path:
    # PEDONI_HIP_LIB: another build of the SAME library (A/B of compiler flags in tools/); a
    # path that does not exist fails loudly like the default one
    import os
    override = os.environ.get("PEDONI_HIP_LIB")
    return Path(override) if override else _ROOT / "lib" / "libpedoni_hip.so"


def _bind(path: Path, mode: int) -> C.CDLL:
    lib = C.CDLL(str(path), mode=mode)
    lib.pedoni_hip_last_error.restype = C.c_char_p
    lib.pedoni_hip_kernel_name.restype = C.c_char_p
    lib.pedoni_hip_kernel_name.argtypes = [C.c_int32]
    lib.pedoni_hip_destroy.restype = None
    lib.pedoni_hip_destroy.argtypes = [C.c_void_p]
    lib.pedoni_hip_default_options.restype = None
    return lib


def load_library() -> C.CDLL:
    """dlopen the in-tree HIP library; never builds, never substitutes."""
    global _LIB
    if _LIB is not None:
        return _LIB
    path = library_path()
    if not path.exists():
        raise PedoniError(
            f"{path} is missing: build it with `python -m pedoni_amd.build` "
            "(the HIP backend has no CPU fallback)")
    _LIB = _bind(path, C.RTLD_GLOBAL)
    return _LIB


_DIAG: Optional[C.CDLL] = None


def diagnostics_library_path() -> Path:
    return _ROOT / "lib" / "libpedoni_hip_diag.so"


def load_diagnostics_library() -> C.CDLL:
    """The -DPEDONI_DIAGNOSTICS build of the same sources (fault-injection hook, instrumented and
    ablation force kernels): for tests and tools only; `HipModel(..., diagnostics=True)` uses it."""
    global _DIAG
    if _DIAG is None:
        path = diagnostics_library_path()
        if not path.exists():
            raise PedoniError(f"{path} is missing: build it with `python -m pedoni_amd.build`")
        _DIAG = _bind(path, C.RTLD_LOCAL)
    return _DIAG


def _check(lib: C.CDLL, rc: int) -> None:
    if rc != 0:
        msg = lib.pedoni_hip_last_error()
        raise PedoniError(f"pedoni_hip error {rc}: {msg.decode() if msg else '?'}")


def _f32(a, shape_last: Optional[int] = None) -> np.ndarray:
    a = np.ascontiguousarray(a, dtype=np.float32)
    if shape_last is not None:
        a = a.reshape(-1, shape_last)
    return a


def _ptr(a: Optional[np.ndarray], typ):
    return None if a is None else a.ctypes.data_as(C.POINTER(typ))


def device_count() -> int:
    lib = load_library()
    n = C.c_int32(0)
    rc = lib.pedoni_hip_device_count(C.byref(n))
    return int(n.value) if rc == 0 else 0


def selftest_math(op: int, a, b=None, math_mode: int = MATH_EXACT, device: int = 0) -> np.ndarray:
    lib = load_library()
    a = _f32(a).ravel()
    bb = None if b is None else _f32(b).ravel()
    out = np.empty_like(a)
    _check(lib, lib.pedoni_hip_selftest_math(
        C.c_int(device), C.c_int32(op), C.c_int32(math_mode), _ptr(a, C.c_float),
        _ptr(bb, C.c_float), _ptr(out, C.c_float), C.c_uint32(a.size)))
    return out


def selftest_field(grid, px, py, device: int = 0):
    """(sobel_filter, bilinear centre) of a map at grid-coordinate points, on the device."""
    lib = load_library()
    g = _f32(grid)
    px, py = _f32(px).ravel(), _f32(py).ravel()
    grad, centre = np.zeros((len(px), 2), np.float32), np.zeros(len(px), np.float32)
    _check(lib, lib.pedoni_hip_selftest_field(
        C.c_int(device), _ptr(g, C.c_float), C.c_uint32(g.shape[0]), C.c_uint32(g.shape[1]),
        _ptr(px, C.c_float), _ptr(py, C.c_float), _ptr(grad, C.c_float), _ptr(centre, C.c_float),
        C.c_uint32(len(px))))
    return grad, centre


def selftest_pair(pos, e, pos_i, vel_i, acc=None, math_mode: int = MATH_EXACT, device: int = 0) -> np.ndarray:
    """acc + pair force (sfm.rs:130-153) of n independent pairs, evaluated on the device."""
    lib = load_library()
    pos, e, pos_i, vel_i = (_f32(a).reshape(-1, 2) for a in (pos, e, pos_i, vel_i))
    out = np.zeros_like(pos) if acc is None else _f32(acc).reshape(-1, 2).copy()
    _check(lib, lib.pedoni_hip_selftest_pair(
        C.c_int(device), C.c_int32(math_mode), _ptr(pos, C.c_float), _ptr(e, C.c_float),
        _ptr(pos_i, C.c_float), _ptr(vel_i, C.c_float), _ptr(out, C.c_float), C.c_uint32(len(pos))))
    return out


class HipModel:
    """One GPU's `PedestrianModel` (models/mod.rs:13-25) behind the C-ABI."""

    def __init__(self, options: Options, size: Sequence[float], distance_map: np.ndarray,
                 potential_maps: Sequence[np.ndarray], field_unit: float,
                 obstacles: Optional[np.ndarray] = None, device: int = 0,
                 map_rows: Optional[Sequence[int]] = None, diagnostics: bool = False):
        """`map_rows` = (begin, end): upload only these texel rows of every (full-size) map --
        one band of a sharded run (pedoni_hip_create_rows; see shard_map_rows).
        `diagnostics`: run on the diagnostics build of the library (debug_* methods)."""
        self._lib = load_diagnostics_library() if diagnostics else load_library()
        self._h = C.c_void_p(None)
        dm = _f32(distance_map)
        if dm.ndim != 2:
            raise PedoniError("distance_map must be 2-D (rows, cols)")
        pms = [_f32(p) for p in potential_maps]
        for p in pms:
            if p.shape != dm.shape:
                raise PedoniError("potential map shape differs from distance map")
        ptrs = (C.POINTER(C.c_float) * max(len(pms), 1))(*[_ptr(p, C.c_float) for p in pms])
        obs = np.zeros((0, 5), np.float32) if obstacles is None else _f32(obstacles, 5)
        self.options = options
        self.n_maps = len(pms)
        opt = options._c()
        r0, r1 = (0, dm.shape[0]) if map_rows is None else (int(map_rows[0]), int(map_rows[1]))
        self.map_rows = (r0, r1)
        rc = self._lib.pedoni_hip_create_rows(
            C.byref(opt), C.c_float(size[0]), C.c_float(size[1]), _ptr(dm, C.c_float), ptrs,
            C.c_uint32(len(pms)), C.c_uint32(dm.shape[0]), C.c_uint32(dm.shape[1]),
            C.c_float(field_unit), obs.ctypes.data_as(C.POINTER(_Obstacle)),
            C.c_uint32(obs.shape[0]), C.c_int(device), C.c_uint32(r0), C.c_uint32(r1), C.byref(self._h))
        _check(self._lib, rc)

    # -- lifetime ----------------------------------------------------------------
    def close(self) -> None:
        if getattr(self, "_h", None) is not None and self._h.value:
            self._lib.pedoni_hip_destroy(self._h)
            self._h = C.c_void_p(None)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    # -- trait methods -----------------------------------------------------------
    def spawn_pedestrians(self, pos=None, destination=None) -> None:
        """PedestrianModel::spawn_pedestrians: append (vel 0, drawn speed) + sort/despawn."""
        n = 0 if pos is None else len(pos)
        peds = np.zeros(n, PED_DTYPE)
        if n:
            p = _f32(pos, 2)
            peds["x"], peds["y"] = p[:, 0], p[:, 1]
            peds["destination"] = np.asarray(destination, dtype=np.uint64)
        _check(self._lib, self._lib.pedoni_hip_spawn_pedestrians(
            self._h, peds.ctypes.data_as(C.POINTER(_Pedestrian)), C.c_uint32(n)))

    def update_states(self) -> None:
        _check(self._lib, self._lib.pedoni_hip_update_states(self._h))

    def list_pedestrians(self) -> np.ndarray:
        n = C.c_uint32(0)
        _check(self._lib, self._lib.pedoni_hip_list_pedestrians(self._h, None, 0, C.byref(n)))
        out = np.zeros(n.value, PED_DTYPE)
        _check(self._lib, self._lib.pedoni_hip_list_pedestrians(
            self._h, out.ctypes.data_as(C.POINTER(_Pedestrian)), C.c_uint32(n.value), C.byref(n)))
        return out

    def get_pedestrian_count(self) -> int:
        c = C.c_int32(0)
        _check(self._lib, self._lib.pedoni_hip_get_pedestrian_count(self._h, C.byref(c)))
        return int(c.value)

    # -- extensions --------------------------------------------------------------
    def append(self, pos, destination, desired_speed=None, vel=None) -> None:
        p = _f32(pos, 2)
        d = np.ascontiguousarray(destination, dtype=np.uint32)
        v0 = None if desired_speed is None else _f32(desired_speed).ravel()
        v = None if vel is None else _f32(vel, 2)
        if len(d) != len(p) or (v0 is not None and len(v0) != len(p)) or \
                (v is not None and len(v) != len(p)):
            raise PedoniError("append: array lengths differ")
        _check(self._lib, self._lib.pedoni_hip_append(
            self._h, _ptr(p, C.c_float), _ptr(d, C.c_uint32), _ptr(v0, C.c_float),
            _ptr(v, C.c_float), C.c_uint32(len(p))))

    def sort_despawn(self) -> None:
        _check(self._lib, self._lib.pedoni_hip_sort_despawn(self._h))

    def tick_n(self, steps: int) -> None:
        _check(self._lib, self._lib.pedoni_hip_tick_n(self._h, C.c_uint32(steps)))

    def tick(self) -> dict:
        m = _StepMetrics()
        _check(self._lib, self._lib.pedoni_hip_tick(self._h, C.byref(m)))
        k = m.time_calc_state_kernel
        return {"active_ped_count": m.active_ped_count, "time_spawn": m.time_spawn,
                "time_calc_state": m.time_calc_state,
                "time_calc_state_kernel": None if k < 0 else k}

    def download(self):
        """(pos[n,2], destination[n], vel[n,2], desired_speed[n]) in model order."""
        n = C.c_uint32(0)
        _check(self._lib, self._lib.pedoni_hip_download(self._h, None, None, None, None, 0,
                                                       C.byref(n)))
        k = n.value
        pos = np.empty((k, 2), np.float32)
        vel = np.empty((k, 2), np.float32)
        v0 = np.empty(k, np.float32)
        dest = np.empty(k, np.uint32)
        _check(self._lib, self._lib.pedoni_hip_download(
            self._h, _ptr(pos, C.c_float), _ptr(dest, C.c_uint32), _ptr(vel, C.c_float),
            _ptr(v0, C.c_float), C.c_uint32(k), C.byref(n)))
        return pos, dest, vel, v0

    def get_spawn_rng(self):
        """(position stream, desired-speed stream) generator states."""
        a, b = C.c_uint64(0), C.c_uint64(0)
        _check(self._lib, self._lib.pedoni_hip_get_spawn_rng(self._h, C.byref(a), C.byref(b)))
        return int(a.value), int(b.value)

    def set_speed_rng(self, state: int) -> None:
        """Restore the desired-speed stream (sfm.rs:54), e.g. from a checkpoint."""
        _check(self._lib, self._lib.pedoni_hip_set_speed_rng(self._h, C.c_uint64(state)))

    def clear(self) -> None:
        _check(self._lib, self._lib.pedoni_hip_clear(self._h))

    def neighbor_grid_shape(self):
        r, c = C.c_uint32(0), C.c_uint32(0)
        _check(self._lib, self._lib.pedoni_hip_neighbor_grid_shape(self._h, C.byref(r), C.byref(c)))
        return int(r.value), int(c.value)

    def cell_flags(self) -> np.ndarray:
        """The per-cell early-out table (include/pedoni_hip.h), shape (rows, cols); empty when left out."""
        n = C.c_uint32(0)
        _check(self._lib, self._lib.pedoni_hip_cell_flags(self._h, None, 0, C.byref(n)))
        out = np.empty(n.value, np.uint32)
        if n.value:
            _check(self._lib, self._lib.pedoni_hip_cell_flags(self._h, _ptr(out, C.c_uint32), C.c_uint32(n.value), C.byref(n)))
            return out.reshape(self.neighbor_grid_shape())
        return out

    def tile_order(self):
        """(order, weights): the force launch's workgroup -> tile map of the last sort pass and the per-tile weights
        the last force launch left (include/pedoni_hip.h); two empty arrays when the plain order is in use."""
        n = C.c_uint32(0)
        _check(self._lib, self._lib.pedoni_hip_tile_order(self._h, None, None, 0, C.byref(n)))
        order, weight = np.empty(n.value, np.uint32), np.empty(n.value, np.uint32)
        if n.value:
            _check(self._lib, self._lib.pedoni_hip_tile_order(self._h, _ptr(order, C.c_uint32), _ptr(weight, C.c_uint32),
                                                               C.c_uint32(n.value), C.byref(n)))
        return order, weight

    def neighbor_grid_indices(self) -> np.ndarray:
        n = C.c_uint32(0)
        _check(self._lib, self._lib.pedoni_hip_neighbor_grid_indices(self._h, None, 0, C.byref(n)))
        out = np.empty(n.value, np.uint32)
        if n.value:
            _check(self._lib, self._lib.pedoni_hip_neighbor_grid_indices(
                self._h, _ptr(out, C.c_uint32), C.c_uint32(n.value), C.byref(n)))
        return out

    def calc_accelerations(self, n: int) -> np.ndarray:
        out = np.empty((n, 2), np.float32)
        _check(self._lib, self._lib.pedoni_hip_calc_accelerations(
            self._h, _ptr(out, C.c_float), C.c_uint32(n)))
        return out

    def set_stream(self, stream_ptr: Optional[int]) -> None:
        """Run on the given hipStream_t (0 = HIP's default stream); None = library stream."""
        _check(self._lib, self._lib.pedoni_hip_set_stream(
            self._h, C.c_void_p(stream_ptr or 0), C.c_int32(1 if stream_ptr is None else 0)))

    def get_stream(self) -> int:
        s = C.c_void_p(None)
        _check(self._lib, self._lib.pedoni_hip_get_stream(self._h, C.byref(s)))
        return int(s.value or 0)

    def synchronize(self) -> None:
        _check(self._lib, self._lib.pedoni_hip_synchronize(self._h))

    def profile(self, enable, kernels: Optional[Sequence[int]] = None, every: int = 1, burst: int = 0) -> None:
        """Time kernel launches with hipEvent pairs: all kernels, or only the PEDONI_K_*
        indices in `kernels` (each pair costs a few microseconds on the stream); inside tick_n
        only every `every`-th tick is timed (the others may replay the captured graph) -- or, with
        `burst`, that many ticks in a row out of every `every`, starting with the next tick."""
        mask = 0
        if enable:
            mask = 0xFF if kernels is None else sum(1 << k for k in kernels)
        if burst:
            _check(self._lib, self._lib.pedoni_hip_profile_burst(self._h, C.c_uint32(max(1, every)), C.c_uint32(burst)))
        else:
            _check(self._lib, self._lib.pedoni_hip_profile_every(self._h, C.c_uint32(max(1, every))))
        _check(self._lib, self._lib.pedoni_hip_profile(self._h, C.c_int32(mask)))

    def kernel_times(self, reset: bool = False) -> dict:
        t = _KernelTimes()
        _check(self._lib, self._lib.pedoni_hip_kernel_times(self._h, C.byref(t), C.c_int32(int(reset))))
        out = {}
        for k in range(N_KERNELS):
            name = self._lib.pedoni_hip_kernel_name(k).decode()
            out[name] = {"total_ms": float(t.total_ms[k]), "launches": int(t.launches[k])}
        return out

    # -- sharding ----------------------------------------------------------------
    def set_band(self, row_begin: int, row_end: int, halo_cap: int) -> None:
        _check(self._lib, self._lib.pedoni_hip_set_band(self._h, C.c_int32(row_begin),
                                                       C.c_int32(row_end), C.c_uint32(halo_cap)))

    @staticmethod
    def halo_bytes(cap_each: int) -> int:
        lib = load_library()
        b = C.c_uint64(0)
        _check(lib, lib.pedoni_hip_halo_bytes(C.c_uint32(cap_each), C.byref(b)))
        return int(b.value)

    def halo_pack(self, send_dev_ptr: int, cap_each: int) -> None:
        _check(self._lib, self._lib.pedoni_hip_halo_pack(self._h, C.c_void_p(send_dev_ptr),
                                                        C.c_uint32(cap_each)))

    def halo_unpack(self, below_dev_ptr: Optional[int], above_dev_ptr: Optional[int],
                    cap_each: int) -> None:
        _check(self._lib, self._lib.pedoni_hip_halo_unpack(
            self._h, C.c_void_p(below_dev_ptr), C.c_void_p(above_dev_ptr), C.c_uint32(cap_each)))

    def halo_tick(self, below_dev_ptr: Optional[int], above_dev_ptr: Optional[int],
                  send_dev_ptr: int, cap_each: int) -> None:
        _check(self._lib, self._lib.pedoni_hip_halo_tick(
            self._h, C.c_void_p(below_dev_ptr), C.c_void_p(above_dev_ptr),
            C.c_void_p(send_dev_ptr), C.c_uint32(cap_each)))

    def halo_tick_begin(self, below_dev_ptr: Optional[int], above_dev_ptr: Optional[int],
                        send_dev_ptr: int, cap_each: int) -> None:
        _check(self._lib, self._lib.pedoni_hip_halo_tick_begin(
            self._h, C.c_void_p(below_dev_ptr), C.c_void_p(above_dev_ptr),
            C.c_void_p(send_dev_ptr), C.c_uint32(cap_each)))

    def halo_tick_end(self) -> None:
        _check(self._lib, self._lib.pedoni_hip_halo_tick_end(self._h))

    def owned_count(self) -> int:
        c = C.c_int32(0)
        _check(self._lib, self._lib.pedoni_hip_owned_count(self._h, C.byref(c)))
        return int(c.value)

    def force_kernel_info(self, n_agents: int):
        """(symbol of the force kernel a whole-array launch over n_agents takes, agents per wave)."""
        buf = C.create_string_buffer(128)
        per = C.c_uint32(0)
        _check(self._lib, self._lib.pedoni_hip_force_kernel_info(self._h, C.c_uint32(n_agents), buf, C.c_uint32(128),
                                                                 C.byref(per)))
        return buf.value.decode(), int(per.value)

    def debug_force_trace(self, reset: bool = True):
        """Per-phase cycle sums of the instrumented force kernel (PEDONI_FORCE_TRACE=1)."""
        out = (C.c_uint64 * 7)()
        _check(self._lib, self._lib.pedoni_hip_debug_force_trace(self._h, out, C.c_int32(int(reset))))
        return [int(x) for x in out]

    def debug_force_trace_raw(self, n_waves: int) -> np.ndarray:
        """(n_waves, 8) u64: per wave the five phase sums, lifetime, launches, start stamp."""
        out = np.zeros((n_waves, 8), np.uint64)
        _check(self._lib, self._lib.pedoni_hip_debug_force_trace_raw(self._h, out.ctypes.data_as(C.POINTER(C.c_uint64)),
                                                                     C.c_uint32(n_waves)))
        return out

    def debug_set_status(self, word: int) -> None:
        """Test hook: overwrite the sticky device status word (0 clears it)."""
        _check(self._lib, self._lib.pedoni_hip_debug_set_status(self._h, C.c_uint32(word)))

    def debug_set_ablate(self, bits: int) -> None:
        """Diagnostics: switch parts of the force kernel off (timing only, results wrong)."""
        _check(self._lib, self._lib.pedoni_hip_debug_set_ablate(self._h, C.c_uint32(bits)))


# -- multi-GPU driver below the C-ABI (pedoni_shard_*) ---------------------------------------
SHARD_ID_BYTES = 128


def shard_unique_id() -> bytes:
    """ncclGetUniqueId on rank 0; hand the 128 bytes to every rank by any channel."""
    lib = load_library()
    buf = (C.c_uint8 * SHARD_ID_BYTES)()
    _check(lib, lib.pedoni_shard_unique_id(buf))
    return bytes(buf)


def shard_map_rows(row_begin: int, row_end: int, slack_rows: int, neighbor_grid_unit: float, field_unit: float,
                   field_rows: int):
    """Texel rows [begin, end) of the field maps a band of grid rows needs (pure host code)."""
    lib = load_library()
    a, b = C.c_uint32(0), C.c_uint32(0)
    _check(lib, lib.pedoni_shard_map_rows(C.c_int32(row_begin), C.c_int32(row_end), C.c_int32(slack_rows),
                                          C.c_float(neighbor_grid_unit), C.c_float(field_unit),
                                          C.c_uint32(field_rows), C.byref(a), C.byref(b)))
    return int(a.value), int(b.value)


def balanced_bounds(row_counts, world: int, min_rows: int = 6) -> list:
    """Row boundaries that give every band about the same number of AGENTS (pure host code)."""
    lib = load_library()
    rc = np.ascontiguousarray(row_counts, np.uint32)
    out = (C.c_int32 * (world + 1))()
    _check(lib, lib.pedoni_shard_balanced_bounds(rc.ctypes.data_as(C.POINTER(C.c_uint32)), C.c_uint32(len(rc)),
                                                 C.c_int32(world), C.c_int32(min_rows), out))
    return list(out)


def recut_bounds(bounds, row_counts, max_shift: int, bulk_cap: int, bounds0=None, map_slack_rows: int = -1) -> list:
    """One step of the periodic re-cut (pure host code; pedoni_shard_recut_bounds)."""
    lib = load_library()
    world = len(bounds) - 1
    b = (C.c_int32 * (world + 1))(*[int(x) for x in bounds])
    b0 = (C.c_int32 * (world + 1))(*[int(x) for x in (bounds0 if bounds0 is not None else bounds)])
    rc = np.ascontiguousarray(row_counts, np.uint32)
    out = (C.c_int32 * (world + 1))()
    _check(lib, lib.pedoni_shard_recut_bounds(b, C.c_int32(world), rc.ctypes.data_as(C.POINTER(C.c_uint32)),
                                              C.c_uint32(len(rc)), C.c_uint32(max_shift), C.c_uint32(bulk_cap), b0,
                                              C.c_int32(map_slack_rows), out))
    return list(out)


class Shard:
    """One rank's band, driven by the library itself: direct RCCL neighbour exchange (or, with
    `unique_id=None`, a member of a local group on one device -- see `local_group_tick_n`)."""

    def __init__(self, model: "HipModel", rank: int, world: int, row_bounds: Sequence[int], halo_cap: int,
                 unique_id: Optional[bytes] = None):
        self._lib = model._lib
        self.model, self.rank, self.world = model, rank, world
        b = (C.c_int32 * (world + 1))(*[int(x) for x in row_bounds])
        idbuf = None
        if unique_id is not None:
            assert len(unique_id) == SHARD_ID_BYTES
            idbuf = (C.c_uint8 * SHARD_ID_BYTES).from_buffer_copy(unique_id)
        h = C.c_void_p(None)
        _check(self._lib, self._lib.pedoni_shard_create(model._h, C.c_int32(rank), C.c_int32(world), idbuf, b,
                                                        C.c_uint32(halo_cap), C.byref(h)))
        self._h = h

    def close(self) -> None:
        if getattr(self, "_h", None):
            self._lib.pedoni_shard_destroy.restype = None
            self._lib.pedoni_shard_destroy(self._h)
            self._h = None

    def begin(self) -> None:
        _check(self._lib, self._lib.pedoni_shard_begin(self._h))

    def tick_n(self, steps: int) -> None:
        _check(self._lib, self._lib.pedoni_shard_tick_n(self._h, C.c_uint32(steps)))

    def selftest(self) -> None:
        _check(self._lib, self._lib.pedoni_shard_selftest(self._h))

    def set_overlap(self, on: bool) -> None:
        """Exchange of the next tick's lists on its own stream, under the interior rows' update."""
        _check(self._lib, self._lib.pedoni_shard_set_overlap(self._h, C.c_int32(1 if on else 0)))

    def tick_forms(self) -> dict:
        """How the ticks so far were run: {'edge_first', 'split', 'plain'} (pedoni_shard_tick_forms)."""
        e, sp, pl = C.c_uint32(0), C.c_uint32(0), C.c_uint32(0)
        _check(self._lib, self._lib.pedoni_shard_tick_forms(self._h, C.byref(e), C.byref(sp), C.byref(pl)))
        return {"edge_first": int(e.value), "split": int(sp.value), "plain": int(pl.value)}

    def set_rebalance(self, every_ticks: int, max_rows_per_step: int = 4, map_slack_rows: int = -1) -> None:
        _check(self._lib, self._lib.pedoni_shard_set_rebalance(self._h, C.c_uint32(every_ticks),
                                                               C.c_uint32(max_rows_per_step),
                                                               C.c_int32(map_slack_rows)))

    def owned_count(self) -> int:
        c = C.c_int32(0)
        _check(self._lib, self._lib.pedoni_shard_owned_count(self._h, C.byref(c)))
        return int(c.value)

    def band(self):
        a, b = C.c_int32(0), C.c_int32(0)
        _check(self._lib, self._lib.pedoni_shard_band(self._h, C.byref(a), C.byref(b)))
        return int(a.value), int(b.value)

    def download_owned(self):
        """Owned agents (rows of the band) of the current sorted order, ghosts stripped."""
        lo, hi = self.band()
        pos, dest, vel, v0 = self.model.download()
        with np.errstate(invalid="ignore"):
            rows = np.trunc(np.nan_to_num(pos[:, 1] / np.float32(self.model.options.neighbor_grid_unit),
                                          nan=-1e9)).astype(np.int64)
            keep = (rows >= lo) & (rows < hi) & ~np.isnan(pos).any(axis=1)
        return pos[keep], dest[keep], vel[keep], v0[keep]


def local_group_tick_n(shards: Sequence[Shard], steps: int) -> None:
    lib = shards[0]._lib
    arr = (C.c_void_p * len(shards))(*[s._h for s in shards])
    _check(lib, lib.pedoni_shard_local_group_tick_n(arr, C.c_uint32(len(shards)), C.c_uint32(steps)))


def eikonal(potential: np.ndarray, slowness, device: int = 0):
    """pedoni_hip_eikonal on a (rows, cols) f32 array (0 on the zero set, >= 1e23 elsewhere);
    `slowness` = per-cell array or a float.  Returns (solution, relaxation launches)."""
    lib = load_library()
    u = np.array(potential, np.float32, copy=True, order="C")
    n = C.c_uint32(0)
    if np.isscalar(slowness):
        fptr, uni = None, float(slowness)
    else:
        f = np.ascontiguousarray(slowness, np.float32)
        assert f.shape == u.shape
        fptr, uni = f.ctypes.data_as(C.POINTER(C.c_float)), 0.0
    _check(lib, lib.pedoni_hip_eikonal(C.c_int(device), u.ctypes.data_as(C.POINTER(C.c_float)), fptr,
                                       C.c_float(uni), C.c_uint32(u.shape[0]), C.c_uint32(u.shape[1]), C.byref(n)))
    return u, int(n.value)
