#!/bin/bash
# The tick's roctx ranges on a rocprofv3 timeline: PEDONI_ROCTX=1 brackets every pass and
# every kernel launch with a named range.  bash tools/roctx_trace.sh TAG  -> profiles/TAG_roctx_ranges.txt
# (--marker-trace + --kernel-trace only: no counters in this run.)
TAG=${1:?tag}
ROOT=$(pwd); OUT=$ROOT/gpurun_out/roctx_$TAG; rm -rf "$OUT"; mkdir -p "$OUT"
export TMPDIR=/tmp PEDONI_ROCTX=1 PEDONI_NO_GRAPH=1
cd /tmp
rocprofv3 --marker-trace --kernel-trace --output-format csv -d "$OUT/t" -- python3 $ROOT/bench.py --steps 6 --warmup 2 --agents-per-gpu 200000 --no-cpu-baseline --no-fast-leg --no-profile > "$OUT/run.log" 2>&1 || echo "rocprofv3 failed"
cd "$ROOT"
python3 - "$OUT" "$TAG" <<'PY'
import collections, csv, glob, sys
out, tag = sys.argv[1], sys.argv[2]
rows = []
for f in glob.glob(f"{out}/t/**/*marker_api_trace.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
names = collections.Counter()
dur = collections.defaultdict(float)
for r in rows:
    n = r.get("Function") or r.get("Name") or r.get("Message") or str(r)
    names[n] += 1
    try:
        dur[n] += (float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) / 1e3
    except Exception:
        pass
lines = [f"# roctx ranges seen by rocprofv3 --marker-trace (bench.py, 8 eager ticks of 2e5 agents, PEDONI_ROCTX=1 PEDONI_NO_GRAPH=1)",
         f"# {len(rows)} range records", "# count  total_us  name"]
for n, c in names.most_common():
    lines.append(f"{c:6d} {dur[n]:10.1f}  {n}")
open(f"profiles/{tag}_roctx_ranges.txt", "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
PY
cp profiles/${TAG}_roctx_ranges.txt gpurun_out/
