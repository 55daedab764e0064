"""One-off C5-scale check on ONE GPU (not part of the suite): the BASELINE C5 configuration --
8e6 agents in a 1000 x 8000 m box, 8 row bands -- with the 8 bands emulated by 8 models on the
same device (the all-gather replaced by handing every band all send buffers), against one
model holding all 8e6 agents.  Merged band state must equal the single model bit for bit.
    python tools/c5_emulation.py [ticks=5] [agents_per_band=1000000]"""
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import torch                                   # noqa: E402
import bench                                   # noqa: E402
import pedoni_amd as hip                       # noqa: E402
from pedoni_amd import host                    # noqa: E402
from pedoni_amd.sharded import ShardedModel    # noqa: E402
from helpers import bit_equal                  # noqa: E402

ticks = int(sys.argv[1]) if len(sys.argv) > 1 else 5
n_per = int(sys.argv[2]) if len(sys.argv) > 2 else 1_000_000
G = 8
W, H = 1000.0, 1000.0 * G
obs, wps = bench.box_geometry(W, H)
t0 = time.time()
field = host.Field.build((W, H), 0.25, obs, wps)
print(f"field {field.shape} built in {time.time() - t0:.1f} s", flush=True)


def make():
    return hip.HipModel(hip.Options(initial_capacity=int(n_per * 1.2)), (W, H), field.distance_map,
                        field.potential_maps, field.unit, obs)


parts = []
for r in range(G):
    p, d, s, v = bench.uniform_crowd(n_per, (12.0, W - 12.0), (r * 1000.0 + 2.0, (r + 1) * 1000.0 - 2.0), 12345 + r)
    v[:, 1] = np.where(np.arange(n_per) % 2 == 0, 1.1, -1.1)     # make agents cross band edges
    parts.append((p, d, s, v))
pos, dest, v0, vel = (np.concatenate([q[k] for q in parts]) for k in range(4))

single = make()
single.append(pos, dest, v0, vel)
single.sort_despawn()
stream = torch.cuda.current_stream().cuda_stream
cap = 4096
words = hip.HipModel.halo_bytes(cap) // 4
sends = [torch.zeros(words, dtype=torch.int32, device="cuda") for _ in range(G)]
bands = []
for r in range(G):
    m = make()
    m.set_stream(stream)
    bands.append(ShardedModel(m, r, G, halo_cap=cap, gather=lambda s, rv: None, send=sends[r], recv=sends))
owner = bands[0].owner_of(pos[:, 1])
for r, b in enumerate(bands):
    sel = owner == r
    b.load(pos[sel], dest[sel], v0[sel], vel[sel])
print("loaded", [int((owner == r).sum()) for r in range(G)], flush=True)

for _ in range(ticks):
    single.update_states()
    single.sort_despawn()
for t in range(ticks + 1):
    for b in bands:
        b.pack()
    for b in bands:
        b.unpack()
        b.model.sort_despawn()
        if t < ticks:
            b.model.update_states()
torch.cuda.synchronize()
want = single.download()
got_parts = [b.download_owned() for b in bands]
got = [np.concatenate([p[k] for p in got_parts]) for k in range(4)]
ok = len(got[0]) == len(want[0]) and np.array_equal(got[1], want[1]) and \
    all(bit_equal(got[k], want[k]).all() for k in (0, 2, 3))
print(f"{len(want[0])} agents after {ticks} ticks; owned per band {[b.owned_count() for b in bands]}; "
      f"bands == single model bit for bit: {ok}", flush=True)
sys.exit(0 if ok else 1)
