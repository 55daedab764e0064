import sys, numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import torch
import pedoni_amd as hip
from pedoni_amd.sharded import ShardedModel
from oracle import pyoracle as oracle
import helpers
from test_gpu_sharded import _tall_box
world, n = 3, 50000
sc = _tall_box(70.0, 210.0)
field = helpers.oracle_field(oracle, sc)
pos, dest, v0, vel = helpers.inject_crowd(field, sc.field.size, n, 2, seed=70 + world)
vel[:, 1] += np.where(np.arange(n) % 2 == 0, 1.2, -1.2).astype(np.float32)
def make():
    return hip.HipModel(hip.Options(), sc.field.size, field.distance_map, field.potential_maps, field.unit, sc.obstacle_array())
single = make(); single.append(pos, dest, v0, vel); single.sort_despawn()
stream = torch.cuda.current_stream().cuda_stream  # 0 = default stream: one order for all bands
models = [make() for _ in range(world)]
cap = 4096
words = hip.HipModel.halo_bytes(cap) // 4
sends = [torch.zeros(words, dtype=torch.int32, device="cuda") for _ in range(world)]
bands = []
for r, m in enumerate(models):
    m.set_stream(stream)
    bands.append(ShardedModel(m, r, world, halo_cap=cap, gather=lambda s, rv: None, send=sends[r], recv=sends))
owner = bands[0].owner_of(pos[:, 1])
for r, b in enumerate(bands):
    sel = owner == r
    b.load(pos[sel], dest[sel], v0[sel], vel[sel])
print("bounds", bands[0].bounds)
for t in range(13):
    for b in bands: b.pack()
    for b in bands:
        b.unpack(); b.model.sort_despawn()
    torch.cuda.synchronize()
    want = single.download()
    parts = [b.download_owned() for b in bands]
    got = [np.concatenate([p[k] for p in parts]) for k in range(4)]
    hdr = [s[:8].cpu().numpy().tolist() + s[4+cap*6:4+cap*6+4].cpu().numpy().tolist() for s in sends]
    ws = set(want[3].view(np.uint32).tolist()); gs = set(got[3].view(np.uint32).tolist())
    miss = ws - gs; extra = gs - ws
    print(f"t={t} single={len(want[0])} bands={len(got[0])} counts={[b.owned_count() for b in bands]} missing={len(miss)} extra={len(extra)} hdr={hdr}")
    if miss:
        idx = [i for i,v in enumerate(want[3].view(np.uint32).tolist()) if v in miss]
        print("  missing pos", want[0][idx], "rows", np.floor(want[0][idx][:,1]/np.float32(1.4)))
        for r in range(world):
            buf = sends[r].cpu().numpy().view(np.uint32)
            for name, off in (("down", 0), ("up", 4 + cap * 6)):
                cnt = buf[off]
                rec = buf[off + 4: off + 4 + cnt * 6].reshape(-1, 6)
                v0s = set(rec[:, 4].tolist())
                print(f"   rank {r} {name}: count {cnt}, contains {len(miss & v0s)} of the missing")
            full = bands[r].model.download()
            print(f"   rank {r} full local download has {len(miss & set(full[3].view(np.uint32).tolist()))} of the missing; n_local={len(full[0])}")
        break
    eq = helpers.bit_equal(got[0], want[0]).all() and helpers.bit_equal(got[2], want[2]).all()
    if not eq:
        bad = np.where(~helpers.bit_equal(got[0], want[0]).all(axis=1) | ~helpers.bit_equal(got[2], want[2]).all(axis=1))[0]
        print("  state differs at", bad[:10], "rows", np.floor(want[0][bad[:10]][:,1]/np.float32(1.4)), want[0][bad[:3]], got[0][bad[:3]], want[2][bad[:3]], got[2][bad[:3]])
        break
    single.update_states(); single.sort_despawn()
    for b in bands: b.model.update_states()
