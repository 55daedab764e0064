#!/usr/bin/env python3
"""Summarise tools/profile_stalls.sh passes into profiles/TAG_stalls.json.

    tools/stall_summary.py TAG gpurun_out/stalls_TAG [SUFFIX]     (SUFFIX: "stalls" by default; "mem" for
                                                                    tools/profile_mem.sh -> profiles/TAG_mem.json)

Per kernel: the average of every collected counter over its launches, plus derived shares of
the wave lifetime (SQ_WAVE_CYCLES, quad-cycles summed over waves): issuing (ACTIVE_INST_ANY),
parked on s_waitcnt / barrier (WAIT_ANY), issue-stalled (WAIT_INST_ANY).
"""
import collections
import csv
import glob
import json
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent


def main():
    tag, out_dir = sys.argv[1], sys.argv[2]
    suffix = sys.argv[3] if len(sys.argv) > 3 else "stalls"
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(f"{out_dir}/**/*_counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            name = r["Kernel_Name"].split("(")[0].replace("void ", "")
            agg[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
    # launch durations from the kernel traces of the same (profiled) passes
    dur = collections.defaultdict(list)
    for f in glob.glob(f"{out_dir}/**/*_kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            name = r["Kernel_Name"].split("(")[0].replace("void ", "")
            dur[name].append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
    res = {}
    for k, cs in agg.items():
        if not k.startswith("pedoni::"):
            continue
        d = {c: sum(v) / len(v) for c, v in cs.items()}
        d["launches_seen"] = max(len(v) for v in cs.values())
        if dur.get(k):
            d["profiled_launch_us"] = sum(dur[k]) / len(dur[k]) / 1e3
            # shader clock = GRBM_GUI_ACTIVE / 8 XCDs / duration (rocprofv3 sums the XCDs; MI355X_MICROARCH.md, DVFS) --
            # only where the launch is long against what the counter also sees around it (its start-up and
            # read-out: several us): for a 5-us scan the quotient came out at 5.7 "GHz".  Short kernels get the
            # guide's nominal 2.4 GHz, and the file says which it is.
            if d.get("GRBM_GUI_ACTIVE") and d["profiled_launch_us"] >= 50.0:
                d["clock_ghz"] = d["GRBM_GUI_ACTIVE"] / 8.0 / (d["profiled_launch_us"] * 1e3)
                d["clock_source"] = "GRBM_GUI_ACTIVE / 8 / profiled launch duration"
            else:
                d["clock_ghz"] = 2.4
                d["clock_source"] = "nominal (launch shorter than 50 us: the counter quotient is not a clock)"
        wc = d.get("SQ_WAVE_CYCLES")
        if wc:
            for c in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU",
                      "SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_SCA", "SQ_ACTIVE_INST_VMEM", "SQ_WAIT_INST_LDS"):
                if c in d:
                    d["share_" + c[3:].lower()] = d[c] / wc
        if d.get("SQ_WAVES") and wc:
            d["quad_cycles_per_wave"] = wc / d["SQ_WAVES"]
        if d.get("SQ_WAVES") and d.get("SQ_INSTS_VALU"):
            d["valu_insts_per_wave"] = d["SQ_INSTS_VALU"] / d["SQ_WAVES"]
        res[k] = d
    p = ROOT / "profiles" / f"{tag}_{suffix}.json"
    p.write_text(json.dumps(res, indent=1, sort_keys=True))
    for k, d in res.items():
        if "force" in k:
            print(k, json.dumps({c: (v if isinstance(v, str) else (round(v, 4) if v < 10 else round(v))) for c, v in sorted(d.items())}, indent=1))
    print("wrote", p)


if __name__ == "__main__":
    main()
