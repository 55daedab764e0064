"""Under rocprofv3 --kernel-trace: 1e6 agents, warm, then eager ticks with place_kernel's diagnostics switch set
(argv[1]: 0 = full kernel, 16 = returns at once, 64 = bare move).  tools/place_probe.sh prints the timeline."""
import os
import sys
os.environ["PEDONI_NO_GRAPH"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from pedoni_amd import abi, host  # noqa: E402
bits = int(sys.argv[1]) if len(sys.argv) > 1 else 0
side = 1000.0
obstacles, waypoints = bench.box_geometry(side, side)
field = host.Field.build((side, side), 0.25, obstacles, waypoints)
pos, dest, v0, vel = bench.uniform_crowd(1_000_000, (12.0, side - 12.0), (2.0, side - 2.0), 12345)
m = abi.HipModel(abi.Options(initial_capacity=1_300_000), (side, side), field.distance_map, field.potential_maps, field.unit,
                 obstacles, diagnostics=True)
m.append(pos, dest, v0, vel)
m.tick_n(20)
m.synchronize()
m.debug_set_ablate(bits << 8)
m.tick_n(6)
m.synchronize()
m.debug_set_ablate(0)
m.close()
