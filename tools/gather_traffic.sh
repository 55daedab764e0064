#!/bin/bash
# FETCH_SIZE against known read traffic: bash tools/gather_traffic.sh TAG -> profiles/TAG_gather_traffic.txt
TAG=${1:?tag}
ROOT=$(pwd); OUT=$ROOT/gpurun_out/gather_$TAG; rm -rf "$OUT"; mkdir -p "$OUT"
hipcc -O3 --offload-arch=gfx950 -o "$OUT/gather_traffic" tools/microbench/gather_traffic.hip || exit 1
export TMPDIR=/tmp; cd /tmp
rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d "$OUT/fetch" -- "$OUT/gather_traffic" > "$OUT/run.log" 2>&1 || echo "pmc pass failed"
cd "$ROOT"
python3 - "$OUT" "$TAG" <<'PY'
import collections, csv, glob, sys
out, tag = sys.argv[1], sys.argv[2]
agg = collections.defaultdict(list)
for f in glob.glob(f"{out}/fetch/**/*_counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == "FETCH_SIZE":
            agg[r["Kernel_Name"].split("(")[0].replace("void ", "")].append(float(r["Counter_Value"]))
dur = collections.defaultdict(list)
for f in glob.glob(f"{out}/fetch/**/*_kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        dur[r["Kernel_Name"].split("(")[0].replace("void ", "")].append((float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) / 1e3)
n = 1 << 24
known = {"stream16": ("16 B/lane stream", 16.0 * n, 16.0 * n), "gather<float>": ("4-B gather", 4.0 * n + 64.0 * n, 4.0 * n + 128.0 * n),
         "gather<HIP_vector_type<float, 2u> >": ("8-B gather", 4.0 * n + 64.0 * n, 4.0 * n + 128.0 * n),
         "gather<HIP_vector_type<float, 4u> >": ("16-B gather", 4.0 * n + 64.0 * n, 4.0 * n + 128.0 * n)}
lines = ["# FETCH_SIZE (KB, rocprofv3 --pmc, summed over the XCDs) against reads of known size on one MI355X;",
         "# 2^24 elements per launch, 1 GiB array (every distinct line from HBM); tools/microbench/gather_traffic.hip",
         "# kernel | what | FETCH_SIZE x 1024 (MB) | known bytes if a miss fetches 64 B / 128 B (MB) | FETCH/known64 | FETCH/known128 | us"]
for k, v in sorted(agg.items()):
    name = next((kk for kk in known if kk in k), None)
    if not name:
        continue
    what, b64, b128 = known[name]
    fetch = sum(v) / len(v) * 1024.0
    d = sum(dur[k]) / max(len(dur[k]), 1)
    lines.append(f"{name:40s} | {what:16s} | {fetch / 1e6:9.1f} | {b64 / 1e6:8.1f} / {b128 / 1e6:8.1f} | {fetch / b64:5.2f} | {fetch / b128:5.2f} | {d:8.1f}")
open(f"profiles/{tag}_gather_traffic.txt", "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
PY
cp profiles/${TAG}_gather_traffic.txt gpurun_out/
