"""Scenario schema and TOML loader (pedoni-simulator/src/scenario.rs:9-66).

Same semantics as serde + toml upstream: unknown keys are ignored (no
`deny_unknown_fields`; scenarios/random.toml:3 sets a `field.unit` nobody reads), integers
coerce to floats, `width` defaults to 1.0 (scenario.rs:3-5,25-26,41-42) and `spawn` is an
internally tagged enum: {kind="periodic", frequency} | {kind="once", count}
(scenario.rs:60-66).
"""
from __future__ import annotations

from dataclasses import dataclass, field as dc_field
from pathlib import Path
from typing import List, Optional, Tuple, Union

import numpy as np

try:  # Python >= 3.11
    import tomllib as _toml
except ModuleNotFoundError:  # pragma: no cover
    import tomli as _toml


@dataclass
class FieldConfig:  # scenario.rs:17-20
    size: Tuple[float, float] = (0.0, 0.0)


@dataclass
class SegmentConfig:  # scenario.rs:22-52 ObstacleConfig / WaypointConfig
    line: Tuple[Tuple[float, float], Tuple[float, float]] = ((0.0, 0.0), (0.0, 0.0))
    width: float = 1.0

    def row(self) -> List[float]:
        return [self.line[0][0], self.line[0][1], self.line[1][0], self.line[1][1], self.width]


@dataclass
class SpawnPeriodic:  # scenario.rs:63
    frequency: float


@dataclass
class SpawnOnce:  # scenario.rs:64
    count: int


@dataclass
class PedestrianConfig:  # scenario.rs:54-58
    origin: int
    destination: int
    spawn: Union[SpawnPeriodic, SpawnOnce]


@dataclass
class Scenario:  # scenario.rs:9-15
    field: FieldConfig = dc_field(default_factory=FieldConfig)
    waypoints: List[SegmentConfig] = dc_field(default_factory=list)
    obstacles: List[SegmentConfig] = dc_field(default_factory=list)
    pedestrians: List[PedestrianConfig] = dc_field(default_factory=list)

    def obstacle_array(self) -> np.ndarray:
        return np.array([o.row() for o in self.obstacles], np.float32).reshape(-1, 5)

    def waypoint_array(self) -> np.ndarray:
        return np.array([w.row() for w in self.waypoints], np.float32).reshape(-1, 5)


def _vec2(v) -> Tuple[float, float]:
    if not (isinstance(v, (list, tuple)) and len(v) == 2):
        raise ValueError(f"expected [x, y], got {v!r}")
    return (float(v[0]), float(v[1]))


def _segment(d: dict) -> SegmentConfig:
    if "line" not in d:
        raise ValueError("missing field `line`")
    l = d["line"]
    if not (isinstance(l, (list, tuple)) and len(l) == 2):
        raise ValueError(f"`line` must hold two points, got {l!r}")
    return SegmentConfig((_vec2(l[0]), _vec2(l[1])), float(d.get("width", 1.0)))


def _spawn(d: dict):
    kind = d.get("kind")
    if kind == "periodic":
        return SpawnPeriodic(float(d["frequency"]))
    if kind == "once":
        c = d["count"]
        if isinstance(c, float):
            raise ValueError("`count` must be an integer")  # serde i32
        return SpawnOnce(int(c))
    raise ValueError(f"unknown variant `{kind}`, expected `periodic` or `once`")


def from_dict(doc: dict) -> Scenario:
    for key in ("field", "waypoints", "obstacles", "pedestrians"):  # no #[serde(default)]
        if key not in doc:
            raise ValueError(f"missing field `{key}`")
    sc = Scenario()
    sc.field = FieldConfig(_vec2(doc["field"]["size"]))
    sc.waypoints = [_segment(w) for w in doc["waypoints"]]
    sc.obstacles = [_segment(o) for o in doc["obstacles"]]
    for p in doc["pedestrians"]:
        sc.pedestrians.append(PedestrianConfig(int(p["origin"]), int(p["destination"]),
                                               _spawn(p["spawn"])))
    return sc


def loads(text: str) -> Scenario:
    return from_dict(_toml.loads(text))


def load(path: Union[str, Path]) -> Scenario:
    return loads(Path(path).read_text())
