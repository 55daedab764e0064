"""What each part of place_kernel costs: a crowd of 1e6 agents is ticked normally, then ONE sort pass
runs with parts of the kernel switched off (diagnostics build; that pass's results are wrong) and
the place launch is timed with hipEvents.  gpurun -- python tools/ablate_place.py [warm ticks]"""
import sys
sys.path.insert(0, ".")
import bench  # noqa: E402
from pedoni_amd import abi, host  # noqa: E402

WARM = int(sys.argv[1]) if len(sys.argv) > 1 else 30
side = 1000.0
obstacles, waypoints = bench.box_geometry(side, side)
field = host.Field.build((side, side), 0.25, obstacles, waypoints)
pos, dest, v0, vel = bench.uniform_crowd(1_000_000, (12.0, side - 12.0), (2.0, side - 2.0), 12345)
NAMES = {0: "full kernel", 1: "no rank scan", 2: "no record move (28 B in, 28 B out)", 4: "no old-range loads (and no scan)",
         3: "no rank scan, no move", 7: "none of the three"}
for bits in (0, 1, 2, 4, 3, 7, 0):
    out = []
    for _ in range(3):
        m = abi.HipModel(abi.Options(initial_capacity=1_300_000), (side, side), field.distance_map, field.potential_maps,
                         field.unit, obstacles, diagnostics=True)
        m.append(pos, dest, v0, vel)
        m.tick_n(WARM)
        m.get_pedestrian_count()
        m.update_states() if False else None
        m.profile(True, kernels=[abi.K_SLOT], every=1)
        m.kernel_times(reset=True)
        m.debug_set_ablate(bits << 8)
        try:
            m.sort_despawn() if False else m.tick_n(1)
            t = m.kernel_times()["slot"]
            out.append(t["total_ms"] / max(t["launches"], 1) * 1e3)
        except Exception as e:   # noqa: BLE001
            out.append(float("nan"))
        m.debug_set_ablate(0)
        m.close()
    print(f"{NAMES[bits]:40s}: place launch {min(out):6.1f} us (min of {[round(x, 1) for x in out]})")
