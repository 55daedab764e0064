import os, sys, time
sys.path.insert(0, '.')
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29544")
os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
import numpy as np, torch, torch.distributed as dist
import bench
from pedoni_amd import abi, host
from pedoni_amd.sharded import ShardedModel
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
stream = torch.cuda.Stream(); torch.cuda.set_stream(stream)
L = 1000.0
obs, wps = bench.box_geometry(L, L)
field = host.Field.build((L, L), 0.25, obs, wps)
for overlap in (True, False):
    model = abi.HipModel(abi.Options(initial_capacity=1300000), (L, L), field.distance_map, field.potential_maps, field.unit, obs)
    model.set_stream(stream.cuda_stream)
    r = ShardedModel(model, 0, 1, dist, torch, expected_row_agents=1400, overlap=overlap)
    pos, dest, v0, vel = bench.uniform_crowd(1_000_000, (12, L - 12), (2, L - 2), 12345)
    r.load(pos, dest, v0, vel)
    r.tick_n(10); torch.cuda.synchronize()
    for n in (200, 20):
        t0 = time.perf_counter(); r.tick_n(n); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
        print(f"overlap={overlap} n={n}: host submit {1e6*(t1-t0)/n:.1f} us/tick, total {1e6*(t2-t0)/n:.1f} us/tick", flush=True)
    model.close()
dist.destroy_process_group()
