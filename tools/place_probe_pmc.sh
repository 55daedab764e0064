#!/bin/bash
# counters of the probes of tools/place_probe.py (switch 7424: place + place(128) + the signature probes)
ROOT=$(pwd); export TMPDIR=/tmp; OUT=$ROOT/gpurun_out/placeprobe_pmc; rm -rf "$OUT"; mkdir -p "$OUT"
(cd /tmp && rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_SMEM SQ_WAIT_ANY SQ_IFETCH SQ_INSTS_SALU SQ_WAIT_INST_ANY -d "$OUT/t" -- python3 $ROOT/tools/place_probe.py ${1:-7424} > "$OUT/run.log" 2>&1)
python3 - "$OUT" <<'PY'
import collections, csv, glob, sys
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/t/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("pedoni::", "")[:30]
        agg[(n, r["Grid_Size"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
for (n, g), cs in sorted(agg.items()):
    if n.startswith(("place", "probe", "scan")):
        print(f"{n:28s} grid {g:>8s} launches {max(len(v) for v in cs.values()):3d} " + "  ".join(f"{c[3:]}={sum(v[-6:]) / len(v[-6:]):.0f}" for c, v in sorted(cs.items())))
PY
