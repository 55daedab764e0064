"""Host-side cost per sharded tick, piece by piece (1-rank RCCL group, short batches so the
launch queue never pushes back)."""
import os, sys, time
sys.path.insert(0, '.')
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29545")
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
model = abi.HipModel(abi.Options(initial_capacity=1300000), (L, L), field.distance_map, field.potential_maps, field.unit, obs)
model.set_stream(stream.cuda_stream)
r = ShardedModel(model, 0, 1, dist, torch, expected_row_agents=1400, overlap=False)
pos, dest, v0, vel = bench.uniform_crowd(1_000_000, (12, L - 12), (2, L - 2), 12345)
r.load(pos, dest, v0, vel)
r.tick_n(10); torch.cuda.synchronize()
n = 20
for rep in range(2):
    t0 = time.perf_counter()
    for _ in range(n): r._gather(r._send, r._recv)
    t1 = time.perf_counter(); torch.cuda.synchronize()
    print(f"gather only: {1e6*(t1-t0)/n:.1f} us/call")
    r.pack()
    t0 = time.perf_counter()
    for _ in range(n): model.halo_tick(None, None, r._send.data_ptr(), r.cap)
    t1 = time.perf_counter(); torch.cuda.synchronize()
    print(f"halo_tick only: {1e6*(t1-t0)/n:.1f} us/call")
    t0 = time.perf_counter()
    model.tick_n(n) if False else None
    t1 = time.perf_counter()
model.close(); dist.destroy_process_group()
