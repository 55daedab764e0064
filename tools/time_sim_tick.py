import sys, time
sys.path.insert(0, '.')
from pathlib import Path
from pedoni_amd import host
text = Path('tests/golden/scenarios/narrow_gap.toml').read_text()
t0 = time.perf_counter(); sim = host.Simulator(host.SimulatorOptions(seed=1), host.Scenario(text)); print("new: %.3f s" % (time.perf_counter() - t0))
for rep in range(3):
    t0 = time.perf_counter()
    for _ in range(50): m = sim.tick()
    dt = (time.perf_counter() - t0) / 50
    print("tick: %.3f ms  (time_spawn %.3f ms, time_calc_state %.3f ms)" % (dt * 1e3, m['time_spawn'] * 1e3, m['time_calc_state'] * 1e3))
t0 = time.perf_counter()
for _ in range(50): p = sim.list_pedestrians()
print("list_pedestrians: %.3f ms" % ((time.perf_counter() - t0) / 50 * 1e3))
