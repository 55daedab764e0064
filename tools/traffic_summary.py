#!/usr/bin/env python3
"""Summary of tools/profile_traffic.sh's passes -> profiles/TAG_traffic.txt + profiles/TAG_pmc_force.json
(the file bench.py's roofline.traffic is read from)."""
import collections
import csv
import glob
import json
import sys
from pathlib import Path

out, tag = sys.argv[1], sys.argv[2]
ROOT = Path(__file__).resolve().parent.parent


def counters(d):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(f"{d}/**/*_counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            agg[r["Kernel_Name"].split("(")[0].replace("void ", "")][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return {k: {c: sum(v) / len(v) for c, v in cs.items()} for k, cs in agg.items()}


def launches(d):
    n = collections.Counter()
    for f in glob.glob(f"{d}/**/*_kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            n[r["Kernel_Name"].split("(")[0].replace("void ", "")] += 1
    return n


def merged(prefix, passes=("p1", "p2", "p3")):
    m = collections.defaultdict(dict)
    for p in passes:
        for k, v in counters(f"{out}/{prefix}_{p}").items():
            m[k].update(v)
    return m


def read_bytes(c):
    n32, n64, n128 = c.get("TCC_EA0_RDREQ_32B_sum", 0.0), c.get("TCC_EA0_RDREQ_64B_sum", 0.0), c.get("TCC_EA0_RDREQ_128B_sum", 0.0)
    return 32.0 * n32 + 64.0 * n64 + 128.0 * n128


def write_bytes(c):
    n, n64 = c.get("TCC_EA0_WRREQ_sum", 0.0), c.get("TCC_EA0_WRREQ_64B_sum", 0.0)
    return 64.0 * n64 + 32.0 * (n - n64)


lines = [f"# {tag}: fabric-side traffic by request size (rocprofv3 --pmc, one MI355X; tools/profile_traffic.sh)",
         "# read bytes = 32 x RDREQ_32B + 64 x RDREQ_64B + 128 x RDREQ_128B; write bytes = 64 x WRREQ_64B + 32 x (WRREQ - WRREQ_64B)", ""]
# ---- calibration -------------------------------------------------------------------------------
cal = merged("cal", ("p1", "p2"))
n = 1 << 24
known = {"stream16": ("16 B/lane stream of 268 MB", 16.0 * n), "gather<float>": ("4-B gathers, 2^24 distinct lines of a 1 GiB array", None),
         "gather<HIP_vector_type<float, 2u> >": ("8-B gathers", None), "gather<HIP_vector_type<float, 4u> >": ("16-B gathers", None)}
lines.append("## calibration (tools/microbench/gather_traffic.hip): kernel | what | RDREQ 32B / 64B / 128B / all (M) | read MB by size | known MB")
for k, c in sorted(cal.items()):
    name = next((kk for kk in known if kk in k), None)
    if not name:
        continue
    what, kb = known[name]
    kb_txt = f"{kb / 1e6:.1f}" if kb else f"{(4.0 * n + 64.0 * n) / 1e6:.1f} (64 B a miss) .. {(4.0 * n + 128.0 * n) / 1e6:.1f} (128 B a miss)"
    lines.append(f"{name:38s} | {what:48s} | {c.get('TCC_EA0_RDREQ_32B_sum', 0) / 1e6:7.2f} / {c.get('TCC_EA0_RDREQ_64B_sum', 0) / 1e6:7.2f} / "
                 f"{c.get('TCC_EA0_RDREQ_128B_sum', 0) / 1e6:7.2f} / {c.get('TCC_EA0_RDREQ_sum', 0) / 1e6:7.2f} | {read_bytes(c) / 1e6:8.1f} | {kb_txt}")
lines.append("")
# ---- product run: every kernel of the tick ---------------------------------------------------------
prod = merged("product")
fs = counters(f"{out}/product_fetch"); ws = counters(f"{out}/product_write")
lines.append("## product run (bench.py --steps 20, C3 unless stated): kernel | read MB (by size) | of it 32B / 64B / 128B requests (M) | write MB | "
             "DRAM-bound read / write requests (M) | L2 hit / miss / req (M) | FETCH_SIZE MB (x1) | WRITE_SIZE MB")
force_row = None
for k, c in sorted(prod.items()):
    if not any(s in k for s in ("force_kernel", "place_kernel", "scan_rows", "key_kernel", "reorder")):
        continue
    f_kb = fs.get(k, {}).get("FETCH_SIZE", float("nan")); w_kb = ws.get(k, {}).get("WRITE_SIZE", float("nan"))
    lines.append(f"{k[:60]:60s} | {read_bytes(c) / 1e6:7.1f} | {c.get('TCC_EA0_RDREQ_32B_sum', 0) / 1e6:6.2f} / {c.get('TCC_EA0_RDREQ_64B_sum', 0) / 1e6:6.2f} / "
                 f"{c.get('TCC_EA0_RDREQ_128B_sum', 0) / 1e6:6.2f} | {write_bytes(c) / 1e6:6.1f} | {c.get('TCC_EA0_RDREQ_DRAM_sum', 0) / 1e6:6.2f} / {c.get('TCC_EA0_WRREQ_DRAM_sum', 0) / 1e6:6.2f} | "
                 f"{c.get('TCC_HIT_sum', 0) / 1e6:7.2f} / {c.get('TCC_MISS_sum', 0) / 1e6:6.2f} / {c.get('TCC_REQ_sum', 0) / 1e6:7.2f} | {f_kb * 1024 / 1e6:7.1f} | {w_kb * 1024 / 1e6:6.1f}")
    if "force_kernel" in k:
        force_row = (k, c, f_kb, w_kb)
lines.append("")
# ---- ablation: where the force kernel's bytes come from -------------------------------------------
names = {256: "nothing switched off", 291: "no field-map sampling at all (goal + wall stencils, despawn sample)", 260: "no pair work (phases 1-3)",
         258: "no wall term", 257: "no goal stencil", 288: "no despawn sample"}
lines.append("## the diagnostics build's force kernel (force_kernel_queue_ablate) with parts switched off: variant | read MB | write MB | read MB saved vs nothing off")
base = None
abl = {}
for v in (256, 291, 260, 258, 257, 288):
    m = merged(f"abl{v}", ("p1", "p2"))
    k = next((kk for kk in m if "force_kernel" in kk), None)
    if not k:
        continue
    rb, wb = read_bytes(m[k]), write_bytes(m[k])
    abl[v] = (rb, wb)
    if v == 256:
        base = rb
    lines.append(f"{names[v]:72s} | {rb / 1e6:7.1f} | {wb / 1e6:6.1f} | {((base - rb) / 1e6 if base else float('nan')):7.1f}")
lines.append("")
if force_row:
    k, c, f_kb, w_kb = force_row
    rb, wb = read_bytes(c), write_bytes(c)
    agents = None
    lines.append(f"## force kernel, product build ({k}): {rb / 1e6:.1f} MB read + {wb / 1e6:.1f} MB written = {(rb + wb) / 1e6:.1f} MB per launch at the L2's fabric side "
                 f"(Infinity-Cache hits included); requests that went on to DRAM: {c.get('TCC_EA0_RDREQ_DRAM_sum', 0) / 1e6:.2f} M reads, {c.get('TCC_EA0_WRREQ_DRAM_sum', 0) / 1e6:.2f} M writes")
    js = {"workload": "c3" if not any(a in tag for a in ("_c4", "_c2")) else tag, "kernel": "force_integrate", "kernel_symbol": k,
          "read_bytes": rb, "write_bytes": wb, "bytes_per_launch": rb + wb,
          "rdreq_32b": c.get("TCC_EA0_RDREQ_32B_sum"), "rdreq_64b": c.get("TCC_EA0_RDREQ_64B_sum"), "rdreq_128b": c.get("TCC_EA0_RDREQ_128B_sum"),
          "rdreq_dram": c.get("TCC_EA0_RDREQ_DRAM_sum"), "wrreq_dram": c.get("TCC_EA0_WRREQ_DRAM_sum"),
          "tcc_hit": c.get("TCC_HIT_sum"), "tcc_miss": c.get("TCC_MISS_sum"), "tcc_req": c.get("TCC_REQ_sum"),
          "fetch_size_kb_raw": f_kb, "write_size_kb": w_kb,
          "method": "read = 32 x RDREQ_32B + 64 x RDREQ_64B + 128 x RDREQ_128B; write = 64 x WRREQ_64B + 32 x (WRREQ - WRREQ_64B); fabric side of the L2",
          "ablation_read_bytes": {names[v]: abl[v][0] for v in abl}}
    (ROOT / "profiles" / f"{tag}_pmc_force.json").write_text(json.dumps(js, indent=1))
txt = "\n".join(lines) + "\n"
(ROOT / "profiles" / f"{tag}_traffic.txt").write_text(txt)
print(txt)
