#!/usr/bin/env python3
"""Summarise rocprofv3 runs of bench.py into profiles/.

    tools/pmc_summary.py TAG STATS_DIR FETCH_DIR WRITE_DIR [SQ_DIR] [BENCH_JSON]

STATS_DIR: `rocprofv3 --kernel-trace --stats`; FETCH_DIR / WRITE_DIR: separate
`--pmc FETCH_SIZE` / `--pmc WRITE_SIZE` passes (TCC slots do not fit both in one pass,
MI355X_MICROARCH.md "rocprofv3 PMC slots").  HBM bytes per launch follow that guide's gfx950
correction: FETCH_SIZE counts 128-B read requests as 64 B, so reads = 2 x FETCH_SIZE KB;
WRITE_SIZE is exact.  Calibration in our own run: scan kernels read 2.045 MB of u32 counts
and report FETCH_SIZE = 1015-1031 KB (x2 = 2.03-2.06 MB), write 4.09 MB and report 3994 KB.
"""
import collections
import csv
import glob
import json
import shutil
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent


def counters(d):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(f"{d}/**/*_counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            name = r["Kernel_Name"].split("(")[0].replace("void ", "")
            agg[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return {k: {c: sum(v) / len(v) for c, v in cs.items()} for k, cs in agg.items()}


def main():
    tag, stats_dir, fetch_dir, write_dir = sys.argv[1:5]
    sq_dir = sys.argv[5] if len(sys.argv) > 5 and sys.argv[5] != "-" else None
    bench_json = sys.argv[6] if len(sys.argv) > 6 else None
    prof = ROOT / "profiles"
    prof.mkdir(exist_ok=True)
    stats = glob.glob(f"{stats_dir}/**/*_kernel_stats.csv", recursive=True)
    if stats:
        shutil.copy(stats[0], prof / f"{tag}_kernel_stats.csv")
    fetch, write = counters(fetch_dir), counters(write_dir)
    sq = counters(sq_dir) if sq_dir else {}
    raw = {k: {**fetch.get(k, {}), **write.get(k, {}), **sq.get(k, {})} for k in set(fetch) | set(write) | set(sq)}
    (prof / f"{tag}_pmc_raw.json").write_text(json.dumps(raw, indent=1, sort_keys=True))
    workload = None
    if bench_json:
        for line in open(bench_json):
            if line.startswith("{"):
                workload = json.loads(line)["config"]["workload"]
    fk = [k for k in raw if "force_kernel" in k]
    if fk:
        f = raw[fk[0]]
        out = {
            "workload": workload, "kernel": "force_integrate", "kernel_symbol": fk[0],
            "fetch_size_kb_raw": f.get("FETCH_SIZE"), "write_size_kb": f.get("WRITE_SIZE"),
            "hbm_bytes_per_launch": (2.0 * f.get("FETCH_SIZE", 0.0) + f.get("WRITE_SIZE", 0.0)) * 1024.0,
            "correction": "reads = 2 x FETCH_SIZE (gfx950: 128-B requests tallied at 64 B); WRITE_SIZE exact",
            "valu_insts_per_launch": f.get("SQ_INSTS_VALU"), "waves_per_launch": f.get("SQ_WAVES"),
        }
        (prof / f"{tag}_pmc_force.json").write_text(json.dumps(out, indent=1))
        print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
