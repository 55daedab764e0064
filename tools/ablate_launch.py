"""What each part of the force kernel costs, launch by launch: a crowd of 1e6 agents is ticked
normally (relaxing it), then ONE tick runs with parts of the force kernel switched off
(pedoni_hip_debug_set_ablate; results of that tick are wrong) and its force launch is timed with
hipEvents.  A fresh model per measurement, on the diagnostics build of the library
(pedoni_amd/lib/libpedoni_hip_diag.so).   gpurun -- python tools/ablate_launch.py [warm ticks [bits,bits,...]]"""
import sys

import numpy as np

sys.path.insert(0, ".")
import bench  # noqa: E402
from pedoni_amd import abi, host  # noqa: E402

WARM = int(sys.argv[1]) if len(sys.argv) > 1 else 60
NAMES = {1: "goal stencil", 2: "wall stencil", 4: "pairs (phases 1-3)", 8: "phase-2 gather", 16: "phase-2 arithmetic",
         32: "despawn sampling (potential map at the new position)", 64: "row counts", 128: "cell and row counts"}
side = 1000.0
obstacles, waypoints = bench.box_geometry(side, side)
field = host.Field.build((side, side), 0.25, obstacles, waypoints)
pos, dest, v0, vel = bench.uniform_crowd(1_000_000, (20.0, side - 20.0), (2.0, side - 2.0), 1)


def one(bits, reps=3):
    out = []
    for _ in range(reps):
        m = abi.HipModel(abi.Options(initial_capacity=1_300_000), (side, side), field.distance_map,
                         field.potential_maps, field.unit, obstacles, diagnostics=True)
        m.append(pos, dest, v0, vel)
        m.tick_n(WARM)
        m.get_pedestrian_count()
        m.profile(True, kernels=[abi.K_FORCE], every=1)
        m.kernel_times(reset=True)
        m.debug_set_ablate(bits)
        try:
            m.tick_n(1)
            t = m.kernel_times()
            fk = next(v for k, v in t.items() if "force" in k)
            out.append(fk["total_ms"] / max(fk["launches"], 1) * 1e3)
        except Exception as e:  # noqa: BLE001 -- the status word of a tick whose counts were switched off
            out.append(float("nan"))
        m.close()
    return float(np.nanmedian(out))


base = one(0)
print(f"force kernel after {WARM} normal ticks: {base:.1f} us")
for bits in [int(x) for x in sys.argv[2].split(',')] if len(sys.argv) > 2 else (1, 2, 3, 4, 7, 8, 16, 24, 32, 64, 128, 32 | 128, 7 | 32, 7 | 64, 7 | 128, 7 | 32 | 128):
    t = one(bits)
    off = " + ".join(NAMES[b] for b in NAMES if bits & b)
    print(f"  without {off}: {t:.1f} us ({t - base:+.1f})")
