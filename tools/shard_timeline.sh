#!/bin/bash
# Kernel timeline of the sharded tick (1 rank, real RCCL self-exchange): plain and overlapped form.
#   bash tools/shard_timeline.sh TAG  -> gpurun_out/TAG_shard_timeline.txt
TAG=${1:?tag}; ROOT=$(pwd); OUT=$ROOT/gpurun_out/shardtl_$TAG; rm -rf "$OUT"; mkdir -p "$OUT"
export TMPDIR=/tmp PEDONI_FORCE_SHARDED=1
cd /tmp
rocprofv3 --kernel-trace --output-format csv -d "$OUT/t" -- python3 $ROOT/bench.py --steps 30 --warmup 4 --no-cpu-baseline --no-fast-leg --no-profile > "$OUT/run.log" 2>&1 || echo "rocprofv3 failed"
cd "$ROOT"
python3 - "$OUT" > gpurun_out/${TAG}_shard_timeline.txt <<'PY'
import csv, glob, sys, collections
rows = []
for f in glob.glob(sys.argv[1] + "/t/**/*kernel_trace.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
name = lambda r: r["Kernel_Name"].split("(")[0].replace("void ", "").replace("pedoni::", "")[:44]
# ticks = runs starting at halo_unpack_kernel
ticks, cur = [], []
for r in rows:
    if name(r).startswith("halo_unpack_kernel") and cur:
        ticks.append(cur); cur = []
    cur.append(r)
ticks.append(cur)
def describe(t):
    t0 = int(t[0]["Start_Timestamp"])
    return [(name(r), (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, r.get("Stream_Id", "?")) for r in t]
forms = collections.defaultdict(list)
for i in range(len(ticks) - 1):
    t = ticks[i]
    n_force = sum(1 for r in t if name(r).startswith("force_kernel"))
    period = (int(ticks[i + 1][0]["Start_Timestamp"]) - int(t[0]["Start_Timestamp"])) / 1e3
    if period < 400:
        edge_first = any("edge_first" in name(r) for r in t)
        forms["overlapped (edge-first launch)" if edge_first else "overlapped" if n_force >= 2 else "plain"].append((period, t))
for form, lst in forms.items():
    per = sorted(p for p, _ in lst)
    print(f"== {form}: {len(lst)} ticks, period median {per[len(per)//2]:.1f} us")
    med = per[len(per) // 2]
    _, t = min(lst, key=lambda pt: abs(pt[0] - med))
    for n, s, d, st in describe(t):
        print(f"   +{s:7.1f} us  {d:6.1f} us  stream {st}  {n}")
PY
cat gpurun_out/${TAG}_shard_timeline.txt
