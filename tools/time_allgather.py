import os, time
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29555")
os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
import torch, torch.distributed as dist
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
s = torch.cuda.Stream(); torch.cuda.set_stream(s)
send = torch.zeros(27664, dtype=torch.int32, device="cuda"); recv = torch.zeros_like(send)
x = torch.zeros(1 << 20, device="cuda")
for name, fn in (("all_gather_into_tensor", lambda: dist.all_gather_into_tensor(recv, send)),
                 ("tiny kernel only", lambda: x.add_(1.0)),
                 ("all_gather + tiny kernel", lambda: (dist.all_gather_into_tensor(recv, send), x.add_(1.0)))):
    for _ in range(20): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(500): fn()
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"{name}: host {1e6*(t1-t0)/500:.1f} us, total {1e6*(t2-t0)/500:.1f} us per call", flush=True)
dist.destroy_process_group()
