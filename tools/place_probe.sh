#!/bin/bash
# timeline of the last eager ticks of tools/place_probe.py for each switch: bash tools/place_probe.sh
ROOT=$(pwd); export TMPDIR=/tmp
for B in ${SWITCHES:-0 16 64 128}; do
  OUT=$ROOT/gpurun_out/placeprobe_$B; rm -rf "$OUT"; mkdir -p "$OUT"
  (cd /tmp && rocprofv3 --kernel-trace --output-format csv -d "$OUT/t" -- python3 $ROOT/tools/place_probe.py $B > "$OUT/run.log" 2>&1)
  python3 - "$OUT" $B <<'PY'
import csv, glob, sys
rows = []
for f in glob.glob(sys.argv[1] + "/t/**/*kernel_trace.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
name = lambda r: r["Kernel_Name"].split("(")[0].replace("void ", "").replace("pedoni::", "")[:36]
last = [r for r in rows[-12:] if name(r).startswith(("scan", "place", "probe"))][-6:]
t0 = int(last[0]["Start_Timestamp"])
print(f"== place switch {sys.argv[2]}")
for r in last:
    print(f"   +{(int(r['Start_Timestamp']) - t0) / 1e3:7.1f} us  {(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3:6.1f} us  {name(r)}")
PY
done
