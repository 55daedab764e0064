"""CPU suite, part 2: the product's host side without a GPU.

* the C-ABI libraries load and export every symbol their headers declare;
* compute entry points fail loudly without a device (no CPU fallback);
* the C++ TOML reader agrees with tomli; the C++ field builder agrees bit for bit with
  the oracle's restatement of field.rs.
"""
import ctypes as C
import re
from pathlib import Path

import numpy as np
import pytest

from helpers import GOLDEN, box_scenario, oracle_field, random_obstacle_scenario
from pedoni_amd import abi, host, scenario as scn

ROOT = Path(__file__).resolve().parent.parent
NO_GPU = abi.device_count() == 0


def _declared(header: str, prefix: str):
    text = (ROOT / "include" / header).read_text()
    return sorted(set(re.findall(rf"\b({prefix}_[a-z_0-9]+)\s*\(", text)))


def test_hip_library_exports_every_declared_symbol():
    lib = abi.load_library()
    declared = _declared("pedoni_hip.h", "pedoni_hip")
    assert declared, "header parse failed"
    assert sorted(abi.SYMBOLS) == declared, "abi.SYMBOLS out of sync with include/pedoni_hip.h"
    for name in declared:
        assert hasattr(lib, name), f"libpedoni_hip.so lacks {name}"


def test_host_library_exports_every_declared_symbol():
    lib = host.load_library()
    declared = sorted(set(_declared("pedoni_host.h", "pedoni_(?:host|scenario|field|simulator)")))
    text = (ROOT / "include" / "pedoni_host.h").read_text()
    declared = sorted(set(re.findall(r"\b(pedoni_(?:host|scenario|field|simulator)_[a-z_0-9]+)\s*\(", text)))
    assert sorted(host.SYMBOLS) == declared
    for name in declared:
        assert hasattr(lib, name), f"libpedoni_host.so lacks {name}"


def test_struct_layouts_match_the_header():
    assert C.sizeof(abi._Options) == 40
    assert C.sizeof(abi._Obstacle) == 20
    assert C.sizeof(abi._Pedestrian) == 16 and abi.PED_DTYPE.itemsize == 16
    assert C.sizeof(abi._StepMetrics) == 32
    assert C.sizeof(abi._KernelTimes) == 16 * abi.N_KERNELS


@pytest.mark.skipif(not NO_GPU, reason="checks the no-device failure mode")
def test_no_device_fails_loudly_no_cpu_fallback():
    dm = np.ones((8, 8), np.float32)
    with pytest.raises(abi.PedoniError, match="no HIP device|no CPU fallback|hip"):
        abi.HipModel(abi.Options(), (2.0, 2.0), dm, [dm], 0.25)
    with pytest.raises(abi.PedoniError):
        abi.selftest_math(0, np.ones(4, np.float32), np.ones(4, np.float32))
    sc = host.Scenario((GOLDEN / "scenarios" / "narrow_gap.toml").read_text())
    with pytest.raises(abi.PedoniError):
        host.Simulator(host.SimulatorOptions(), sc)


def test_reference_backends_are_not_silently_substituted():
    sc = host.Scenario((GOLDEN / "scenarios" / "narrow_gap.toml").read_text())
    for backend in (host.BACKEND_CPU, host.BACKEND_GPU):
        with pytest.raises(abi.PedoniError, match="not part of this build"):
            host.Simulator(host.SimulatorOptions(backend=backend), sc)


# ---- scenario.rs ---------------------------------------------------------------------------
def _scenario_files():
    files = sorted((GOLDEN / "scenarios").glob("*.toml"))
    ref = Path("/root/reference/scenarios")  # present in the build container only
    if ref.is_dir():
        files += sorted(ref.glob("*.toml"))
    return files


@pytest.mark.parametrize("path", _scenario_files(), ids=lambda p: p.parent.name[:3] + "/" + p.name)
def test_cpp_toml_reader_agrees_with_tomli(path):
    text = path.read_text()
    want = scn.loads(text)
    got = host.Scenario(text)
    assert got.size == tuple(np.float32(v) for v in want.field.size)
    assert np.array_equal(got.waypoints, want.waypoint_array())
    assert np.array_equal(got.obstacles, want.obstacle_array())
    assert len(got.pedestrians) == len(want.pedestrians)
    for g, w in zip(got.pedestrians, want.pedestrians):
        assert (g["origin"], g["destination"]) == (w.origin, w.destination)
        if isinstance(w.spawn, scn.SpawnOnce):
            assert g["spawn"] == {"kind": "once", "count": w.spawn.count}
        else:
            assert g["spawn"] == {"kind": "periodic", "frequency": w.spawn.frequency}


def test_toml_forms_the_reference_files_use():
    text = '''
# comment
[field]
size = [200, 200]   # integers coerce to floats
unit = 0.25         # unknown key: ignored (random.toml:3)

[[waypoints]]
line = [[10, 20], [20, 10]]

[[obstacles]]
line = [
    [
        14.668489878221047,
        138.49110461864066,
    ],
    [ 15.4707104042376, 133.55587998062896, ],
]
width = 0.2

[[obstacles]]
line = [[1e1, -2.5e-1], [+3.0, 4_000]]

[[pedestrians]]
origin = 0
destination = 0
spawn = { kind = "periodic", frequency = 10 }

[[pedestrians]]
origin = 0
destination = 0
spawn = { kind = "once", count = 50 }
'''
    got, want = host.Scenario(text), scn.loads(text)
    assert np.array_equal(got.obstacles, want.obstacle_array())
    assert got.obstacles[0, 4] == np.float32(0.2) and got.obstacles[1, 4] == np.float32(1.0)
    assert got.pedestrians[0]["spawn"] == {"kind": "periodic", "frequency": 10.0}
    assert got.pedestrians[1]["spawn"] == {"kind": "once", "count": 50}


@pytest.mark.parametrize("text,err", [
    ("[field]\nsize=[1,2]\n", "missing field `waypoints`"),
    ("[field]\nsize=[1]\n[[waypoints]]\nline=[[0,0],[1,1]]\n", r"\[x, y\]"),
    ("waypoints=[]\nobstacles=[]\npedestrians=[]\n[field]\nsize=[1,2]\nsize=[3,4]\n", "duplicate key"),
    ("waypoints=[]\nobstacles=[]\n[field]\nsize=[1,2]\n[[pedestrians]]\norigin=0\ndestination=0\n"
     "spawn={kind=\"sometimes\"}\n", "unknown variant"),
    ("waypoints=[]\nobstacles=[]\n[field]\nsize=[1,2]\n[[pedestrians]]\norigin=0\ndestination=0\n"
     "spawn={kind=\"once\", count=1.5}\n", "integer"),
    ("[field\nsize=[1,2]\n", "expected"),
])
def test_toml_errors_are_reported(text, err):
    with pytest.raises(abi.PedoniError, match=err):
        host.Scenario(text)


# ---- field.rs --------------------------------------------------------------------------------
@pytest.mark.parametrize("name", ["narrow_gap", "box", "random"])
def test_cpp_field_builder_equals_oracle_restatement(oracle, name):
    if name == "narrow_gap":
        sc = scn.load(GOLDEN / "scenarios" / "narrow_gap.toml")
    elif name == "box":
        sc = box_scenario(50.0)
    else:
        sc = random_obstacle_scenario(100.0, 150)
    want = oracle_field(oracle, sc)
    got = host.Field.build(sc.field.size, 0.25, sc.obstacle_array(), sc.waypoint_array())
    assert got.shape == want.shape and got.n_maps == len(want.potential_maps)
    assert np.array_equal(got.obstacle_exist, want.obstacle_exist)
    assert np.array_equal(got.distance_map.view(np.uint32), want.distance_map.view(np.uint32))
    for g, w in zip(got.potential_maps, want.potential_maps):
        assert np.array_equal(g.view(np.uint32), w.view(np.uint32))
    p = (sc.field.size[0] * 0.4, sc.field.size[1] * 0.55)
    assert got.get_potential(0, p) == want.get_potential(0, p)
    assert got.get_obstacle_distance(p) == want.get_obstacle_distance(p)


def test_field_shape_is_ceil_of_size_over_unit():
    f = host.Field.build((5.1, 3.0), 0.25, np.zeros((0, 5)), [[1, 1, 1, 2, 1.0]])
    assert f.shape == (12, 21)
