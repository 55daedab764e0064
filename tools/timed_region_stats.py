#!/usr/bin/env python3
"""The launches of bench.py's TIMED REGION in a rocprofv3 kernel trace of that same run.

    tools/timed_region_stats.py TAG STATS_DIR BENCH_JSON_OF_THE_TRACED_RUN

`rocprofv3 --stats` averages every launch of the process: the untimed settling ticks and the warm-up
(a younger, costlier crowd), the timed region, the per-kernel pass after it.  The roofline of the bench
line is the force kernel's average over the timed region only, so this picks exactly those launches out
of the trace (tick index = position among the force kernel's launches: settle + warm-up ticks first,
then `steps` timed ones) and prints them beside the figure the run itself reported from its hipEvents.
-> profiles/TAG_timed_region.txt"""
import csv
import glob
import json
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent


def main():
    tag, stats_dir, bench_json = sys.argv[1:4]
    line = [ln for ln in open(bench_json) if ln.startswith("{")][-1]
    d = json.loads(line)
    first = d["config"]["crowd_age_at_warmup_ticks"] + d["warmup"]
    steps = d["steps"]
    rows = []
    for f in glob.glob(f"{stats_dir}/**/*_kernel_trace.csv", recursive=True):
        rows += list(csv.DictReader(open(f)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    force = [r for r in rows if "force_kernel" in r["Kernel_Name"]]
    if len(force) < first + steps:
        sys.exit(f"only {len(force)} force launches in the trace, need {first + steps}")
    t_begin = int(force[first]["Start_Timestamp"])
    t_end = int(force[first + steps - 1]["End_Timestamp"])
    # the region starts with the sort pass of its first tick: the launches after the previous force kernel
    t_prev = int(force[first - 1]["End_Timestamp"]) if first else 0
    region = [r for r in rows if t_prev < int(r["Start_Timestamp"]) and int(r["End_Timestamp"]) <= t_end
              and "pedoni::" in r["Kernel_Name"]]     # (not the count read-back in front of the region)
    by = {}
    for r in region:
        name = r["Kernel_Name"].split("(")[0].replace("void ", "")
        by.setdefault(name, []).append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    out = [f"# {tag}: launches of bench.py's timed region (ticks {first} .. {first + steps - 1} of the run) in the",
           "# rocprofv3 --kernel-trace of the same process; durations in us",
           f"# command: bench.py --gpus 1 --steps {steps} --warmup {d['warmup']}   (under rocprofv3 --kernel-trace --stats)",
           f"{'kernel':60s} {'launches':>8s} {'avg':>9s} {'min':>9s} {'max':>9s}"]
    for name, v in sorted(by.items(), key=lambda kv: -sum(kv[1])):
        out.append(f"{name[:60]:60s} {len(v):8d} {sum(v) / len(v) / 1e3:9.2f} {min(v) / 1e3:9.2f} {max(v) / 1e3:9.2f}")
    span = (t_end - int(region[0]["Start_Timestamp"])) / 1e3
    out.append(f"first launch to last end: {span:.1f} us = {span / steps:.2f} us per tick (traced: every launch carries the tracer's cost)")
    r = d["roofline"]
    out.append(f"the run's own line: ms_per_step {d['ms_per_step'] * 1e3:.2f} us, roofline.avg_launch_ms "
               f"{r['avg_launch_ms'] * 1e3:.2f} us over {r['timed_launches']} event-timed launches of {r['kernel_symbol']}")
    text = "\n".join(out) + "\n"
    (ROOT / "profiles" / f"{tag}_timed_region.txt").write_text(text)
    print(text)


if __name__ == "__main__":
    main()
