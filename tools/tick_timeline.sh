#!/bin/bash
# Kernel timeline of the timed region of a bench run (start / end of every launch, gaps between them):
#   bash tools/tick_timeline.sh TAG [bench args]   -> gpurun_out/TAG_tick_timeline.txt
TAG=${1:?tag}; shift
ROOT=$(pwd); OUT=$ROOT/gpurun_out/timeline_$TAG; rm -rf "$OUT"; mkdir -p "$OUT"
export TMPDIR=/tmp; cd /tmp
rocprofv3 --kernel-trace --output-format csv -d "$OUT/trace" -- python3 $ROOT/bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-fast-leg --no-profile "$@" > "$OUT/bench.json" 2> "$OUT/bench.err"
cd "$ROOT"
python3 - "$OUT" "$TAG" <<'PY'
import csv, glob, json, sys, statistics
out, tag = sys.argv[1], sys.argv[2]
rows = []
for f in glob.glob(f"{out}/trace/**/*_kernel_trace.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
short = lambda n: n.split("(")[0].replace("void ", "").replace("pedoni::", "")
d = json.loads([l for l in open(f"{out}/bench.json") if l.startswith("{")][-1])
# the timed region = the last 40 ticks' worth of force launches before the end (no per-kernel pass with --no-profile)
force = [i for i, r in enumerate(rows) if "force_kernel" in r["Kernel_Name"]]
first = force[-40]
# start from the first launch of that tick (walk back over scan / place)
while first > 0 and "force_kernel" not in rows[first - 1]["Kernel_Name"]:
    first -= 1
sel = rows[first:force[-1] + 1]
t0 = int(sel[0]["Start_Timestamp"])
dur, gap = {}, {}
prev_end, prev_name = None, None
for r in sel:
    n = short(r["Kernel_Name"]); s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    dur.setdefault(n, []).append((e - s) / 1e3)
    if prev_end is not None:
        gap.setdefault(f"{prev_name} -> {n}", []).append((s - prev_end) / 1e3)
    prev_end, prev_name = e, n
total = (int(sel[-1]["End_Timestamp"]) - t0) / 1e3
lines = [f"# {tag}: kernel timeline of the last 40 ticks of `bench.py --steps 40 {' '.join(sys.argv[3:])}` under rocprofv3 --kernel-trace",
         f"# {total / 40:.2f} us per tick first start -> last end (the run's own line: {d['ms_per_step'] * 1e3:.2f} us per tick)", "kernel | launches | mean us | min | max"]
for n, v in dur.items():
    lines.append(f"{n[:70]:70s} | {len(v):3d} | {statistics.mean(v):7.2f} | {min(v):7.2f} | {max(v):7.2f}")
lines.append("gap (end of one launch -> start of the next) | count | mean us | min | max")
for n, v in gap.items():
    lines.append(f"{n[:110]:110s} | {len(v):3d} | {statistics.mean(v):6.2f} | {min(v):6.2f} | {max(v):6.2f}")
txt = "\n".join(lines) + "\n"
open(f"gpurun_out/{tag}_tick_timeline.txt", "w").write(txt)
print(txt)
PY
