"""ctypes binding of include/pedoni_host.h -- the C++ mirror of pedoni-simulator's
`Simulator`, `Scenario` and `Field` (lib.rs, scenario.rs, field.rs)."""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass
from pathlib import Path
from typing import List, Optional, Union

import numpy as np

from . import abi
from .abi import PedoniError

_LIB: Optional[C.CDLL] = None

BACKEND_CPU, BACKEND_GPU, BACKEND_HIP = 0, 1, 2

SYMBOLS = [
    "pedoni_host_last_error", "pedoni_simulator_default_options", "pedoni_scenario_parse",
    "pedoni_scenario_free", "pedoni_scenario_size", "pedoni_scenario_segments",
    "pedoni_scenario_pedestrians", "pedoni_field_from_scenario", "pedoni_field_build",
    "pedoni_field_build_gpu", "pedoni_field_free", "pedoni_field_shape", "pedoni_field_distance_map",
    "pedoni_field_potential_map", "pedoni_field_obstacle_exist", "pedoni_field_get_potential",
    "pedoni_field_get_obstacle_distance", "pedoni_simulator_new", "pedoni_simulator_free",
    "pedoni_simulator_tick", "pedoni_simulator_tick_n", "pedoni_simulator_step", "pedoni_simulator_list_pedestrians",
    "pedoni_simulator_save_checkpoint", "pedoni_simulator_resume",
    "pedoni_simulator_model", "pedoni_simulator_field",
]


class _SimOptions(C.Structure):
    _fields_ = [("backend", C.c_int32), ("neighbor_grid_unit", C.c_float),
                ("field_grid_unit", C.c_float), ("use_neighbor_grid", C.c_int32),
                ("use_distance_map", C.c_int32), ("gpu_work_size", C.c_int32),
                ("math_mode", C.c_int32), ("device", C.c_int32), ("seed", C.c_uint64)]


@dataclass
class SimulatorOptions:
    """lib.rs:108-135 (same defaults; backend defaults to the variant this build adds)."""
    backend: int = BACKEND_HIP
    neighbor_grid_unit: float = 1.4
    field_grid_unit: float = 0.25
    use_neighbor_grid: bool = True
    use_distance_map: bool = True
    gpu_work_size: int = 64
    math_mode: int = abi.MATH_EXACT
    device: int = 0
    seed: int = 12345

    def _c(self) -> _SimOptions:
        return _SimOptions(self.backend, self.neighbor_grid_unit, self.field_grid_unit,
                           int(self.use_neighbor_grid), int(self.use_distance_map),
                           self.gpu_work_size, self.math_mode, self.device, self.seed)


def library_path() -> Path:
    # PEDONI_HOST_LIB: another build of the SAME library (the sanitizer build of `make asan`)
    import os
    override = os.environ.get("PEDONI_HOST_LIB")
    return Path(override) if override else abi.library_path().with_name("libpedoni_host.so")


def load_library() -> C.CDLL:
    global _LIB
    if _LIB is not None:
        return _LIB
    abi.load_library()  # libpedoni_host.so links against libpedoni_hip.so
    path = library_path()
    if not path.exists():
        raise PedoniError(f"{path} is missing: build it with `python -m pedoni_amd.build`")
    lib = C.CDLL(str(path))
    lib.pedoni_host_last_error.restype = C.c_char_p
    for name in ("pedoni_scenario_free", "pedoni_field_free", "pedoni_simulator_free"):
        getattr(lib, name).restype = None
        getattr(lib, name).argtypes = [C.c_void_p]
    lib.pedoni_field_distance_map.restype = C.POINTER(C.c_float)
    lib.pedoni_field_distance_map.argtypes = [C.c_void_p]
    lib.pedoni_field_potential_map.restype = C.POINTER(C.c_float)
    lib.pedoni_field_potential_map.argtypes = [C.c_void_p, C.c_uint32]
    lib.pedoni_field_obstacle_exist.restype = C.POINTER(C.c_uint8)
    lib.pedoni_field_obstacle_exist.argtypes = [C.c_void_p]
    lib.pedoni_simulator_model.restype = C.c_void_p
    lib.pedoni_simulator_model.argtypes = [C.c_void_p]
    lib.pedoni_simulator_field.restype = C.c_void_p
    lib.pedoni_simulator_field.argtypes = [C.c_void_p]
    _LIB = lib
    return lib


def _check(rc: int) -> None:
    if rc != 0:
        msg = load_library().pedoni_host_last_error()
        raise PedoniError(f"pedoni_host error {rc}: {msg.decode() if msg else '?'}")


class Scenario:
    """scenario.rs:9-15, parsed by the C++ host's TOML reader."""

    def __init__(self, toml_text: str):
        self._lib = load_library()
        self._h = C.c_void_p(None)
        _check(self._lib.pedoni_scenario_parse(toml_text.encode(), C.byref(self._h)))

    @classmethod
    def load(cls, path: Union[str, Path]) -> "Scenario":
        return cls(Path(path).read_text())

    def __del__(self):
        try:
            if self._h:
                self._lib.pedoni_scenario_free(self._h)
                self._h = None
        except Exception:
            pass

    @property
    def size(self):
        out = (C.c_float * 2)()
        _check(self._lib.pedoni_scenario_size(self._h, out))
        return (float(out[0]), float(out[1]))

    def _segments(self, kind: int) -> np.ndarray:
        n = C.c_uint32(0)
        _check(self._lib.pedoni_scenario_segments(self._h, kind, None, 0, C.byref(n)))
        out = np.zeros((n.value, 5), np.float32)
        _check(self._lib.pedoni_scenario_segments(
            self._h, kind, out.ctypes.data_as(C.POINTER(C.c_float)), n.value, C.byref(n)))
        return out

    @property
    def waypoints(self) -> np.ndarray:
        return self._segments(0)

    @property
    def obstacles(self) -> np.ndarray:
        return self._segments(1)

    @property
    def pedestrians(self) -> List[dict]:
        n = C.c_uint32(0)
        _check(self._lib.pedoni_scenario_pedestrians(self._h, None, 0, C.byref(n)))
        raw = np.zeros((n.value, 4), np.float64)
        _check(self._lib.pedoni_scenario_pedestrians(
            self._h, raw.ctypes.data_as(C.POINTER(C.c_double)), n.value, C.byref(n)))
        out = []
        for o, d, k, v in raw:
            spawn = {"kind": "once", "count": int(v)} if k == 1.0 else \
                    {"kind": "periodic", "frequency": float(v)}
            out.append({"origin": int(o), "destination": int(d), "spawn": spawn})
        return out


class Field:
    """field.rs:194-205 built by the C++ host (FieldBuilder + fast marching)."""

    def __init__(self, handle: C.c_void_p, owned: bool = True, keepalive=None):
        self._lib = load_library()
        self._h = handle
        self._owned = owned
        self._keepalive = keepalive
        r, c, m, u = C.c_uint32(0), C.c_uint32(0), C.c_uint32(0), C.c_float(0)
        _check(self._lib.pedoni_field_shape(self._h, C.byref(r), C.byref(c), C.byref(m), C.byref(u)))
        self.shape = (int(r.value), int(c.value))
        self.n_maps = int(m.value)
        self.unit = float(u.value)

    @classmethod
    def from_scenario(cls, scenario: Scenario, unit: float = 0.25) -> "Field":
        lib = load_library()
        h = C.c_void_p(None)
        _check(lib.pedoni_field_from_scenario(scenario._h, C.c_float(unit), C.byref(h)))
        return cls(h)

    @classmethod
    def build(cls, size, unit: float, obstacles, waypoints, solver: str = "heap", device: int = 0) -> "Field":
        """solver="heap": upstream's fast marching, pop order and all (field.rs:118-192) -- the
        parity path.  solver="gpu": opt-in parallel eikonal solver (pedoni_field_build_gpu):
        the scheme's fixed point, NOT upstream's numbers; `.gpu_launches` holds its launch count."""
        lib = load_library()
        obs = np.ascontiguousarray(obstacles, np.float32).reshape(-1, 5)
        wps = np.ascontiguousarray(waypoints, np.float32).reshape(-1, 5)
        h = C.c_void_p(None)
        if solver == "gpu":
            n = C.c_uint32(0)
            _check(lib.pedoni_field_build_gpu(
                C.c_float(size[0]), C.c_float(size[1]), C.c_float(unit),
                obs.ctypes.data_as(C.c_void_p), C.c_uint32(len(obs)),
                wps.ctypes.data_as(C.c_void_p), C.c_uint32(len(wps)), C.c_int32(device), C.byref(n), C.byref(h)))
            f = cls(h)
            f.gpu_launches = int(n.value)
            return f
        if solver != "heap":
            raise PedoniError(f"unknown field solver {solver!r}")
        _check(lib.pedoni_field_build(
            C.c_float(size[0]), C.c_float(size[1]), C.c_float(unit),
            obs.ctypes.data_as(C.c_void_p), C.c_uint32(len(obs)),
            wps.ctypes.data_as(C.c_void_p), C.c_uint32(len(wps)), C.byref(h)))
        return cls(h)

    def __del__(self):
        try:
            if self._owned and self._h:
                self._lib.pedoni_field_free(self._h)
                self._h = None
        except Exception:
            pass

    def _view(self, ptr, dtype) -> np.ndarray:
        n = self.shape[0] * self.shape[1]
        return np.ctypeslib.as_array(ptr, shape=(n,)).view(dtype).reshape(self.shape)

    @property
    def distance_map(self) -> np.ndarray:
        return self._view(self._lib.pedoni_field_distance_map(self._h), np.float32)

    @property
    def potential_maps(self) -> List[np.ndarray]:
        return [self._view(self._lib.pedoni_field_potential_map(self._h, w), np.float32)
                for w in range(self.n_maps)]

    @property
    def obstacle_exist(self) -> np.ndarray:
        return self._view(self._lib.pedoni_field_obstacle_exist(self._h), np.uint8).astype(bool)

    def get_potential(self, waypoint: int, pos) -> float:
        out = C.c_float(0)
        _check(self._lib.pedoni_field_get_potential(self._h, C.c_uint32(waypoint),
                                                    C.c_float(pos[0]), C.c_float(pos[1]), C.byref(out)))
        return float(out.value)

    def get_obstacle_distance(self, pos) -> float:
        out = C.c_float(0)
        _check(self._lib.pedoni_field_get_obstacle_distance(self._h, C.c_float(pos[0]),
                                                            C.c_float(pos[1]), C.byref(out)))
        return float(out.value)


class Simulator:
    """lib.rs:17-105 `Simulator` (new / tick / list_pedestrians, field `step`)."""

    def __init__(self, options: SimulatorOptions, scenario: Scenario, _resume_from=None):
        self._lib = load_library()
        self._h = C.c_void_p(None)
        self.options = options
        self.scenario = scenario
        opt = options._c()
        if _resume_from is None:
            _check(self._lib.pedoni_simulator_new(C.byref(opt), scenario._h, C.byref(self._h)))
        else:
            _check(self._lib.pedoni_simulator_resume(C.byref(opt), scenario._h,
                                                     str(_resume_from).encode(), C.byref(self._h)))

    @classmethod
    def resume(cls, options: SimulatorOptions, scenario: Scenario, path) -> "Simulator":
        """Continue from a file written by `save_checkpoint` (build-owned; upstream has no
        checkpointing): same step counter, generator states and full agent state, so the run
        goes on bit for bit like the uninterrupted one."""
        return cls(options, scenario, _resume_from=path)

    def save_checkpoint(self, path) -> None:
        _check(self._lib.pedoni_simulator_save_checkpoint(self._h, str(path).encode()))

    def close(self) -> None:
        if getattr(self, "_h", None) is not None and self._h.value:
            self._lib.pedoni_simulator_free(self._h)
            self._h = C.c_void_p(None)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def tick(self) -> dict:
        m = abi._StepMetrics()
        _check(self._lib.pedoni_simulator_tick(self._h, C.byref(m)))
        k = m.time_calc_state_kernel
        return {"active_ped_count": m.active_ped_count, "time_spawn": m.time_spawn,
                "time_calc_state": m.time_calc_state,
                "time_calc_state_kernel": None if k < 0 else k}

    def tick_n(self, n: int) -> dict:
        """`n` ticks with the periodic spawners evaluated on the device."""
        m = abi._StepMetrics()
        _check(self._lib.pedoni_simulator_tick_n(self._h, C.c_uint32(n), C.byref(m)))
        return {"active_ped_count": m.active_ped_count, "time_spawn": m.time_spawn,
                "time_calc_state": m.time_calc_state, "time_calc_state_kernel": None}

    @property
    def step(self) -> int:
        s = C.c_int32(0)
        _check(self._lib.pedoni_simulator_step(self._h, C.byref(s)))
        return int(s.value)

    def list_pedestrians(self) -> np.ndarray:
        n = C.c_uint32(0)
        _check(self._lib.pedoni_simulator_list_pedestrians(self._h, None, 0, C.byref(n)))
        out = np.zeros(n.value, abi.PED_DTYPE)
        _check(self._lib.pedoni_simulator_list_pedestrians(
            self._h, out.ctypes.data_as(C.c_void_p), C.c_uint32(n.value), C.byref(n)))
        return out

    @property
    def field(self) -> Field:
        return Field(C.c_void_p(self._lib.pedoni_simulator_field(self._h)), owned=False,
                     keepalive=self)

    @property
    def model_handle(self) -> int:
        return int(self._lib.pedoni_simulator_model(self._h) or 0)
