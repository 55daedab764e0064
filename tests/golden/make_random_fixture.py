#!/usr/bin/env python3
"""Regenerates tests/golden/scenarios/random.toml -- an input-data fixture: the geometry and
spawners of the reference's scenarios/random.toml (BASELINE.json configs[1], C2), one obstacle
per line with every coordinate at its full decimal precision.  Run in the build container
(the GPU box has no /root/reference):  python tests/golden/make_random_fixture.py"""
from pathlib import Path

import tomli

SRC = Path("/root/reference/scenarios/random.toml")
DST = Path(__file__).resolve().parent / "scenarios" / "random.toml"


def num(x):
    return repr(x)           # shortest repr that round-trips the f64 the TOML holds


def main():
    d = tomli.loads(SRC.read_text())
    out = ["# Input fixture (data, not code): field, 4 waypoints, 1004 obstacles and 4 periodic spawners of",
           "# the reference's scenarios/random.toml (BASELINE.json configs[1]), restated compactly for the",
           "# GPU box, where /root/reference is absent.  Regenerate: tests/golden/make_random_fixture.py",
           "[field]", f"size = [{num(d['field']['size'][0])}, {num(d['field']['size'][1])}]",
           f"unit = {num(d['field']['unit'])}", ""]
    for w in d["waypoints"]:
        (a, b), (c, e) = w["line"]
        out += ["[[waypoints]]", f"line = [[{num(a)}, {num(b)}], [{num(c)}, {num(e)}]]"]
        if "width" in w:
            out.append(f"width = {num(w['width'])}")
    out.append("")
    for o in d["obstacles"]:
        (a, b), (c, e) = o["line"]
        out += ["[[obstacles]]", f"line = [[{num(a)}, {num(b)}], [{num(c)}, {num(e)}]]", f"width = {num(o['width'])}"]
    out.append("")
    for p in d["pedestrians"]:
        s = p["spawn"]
        extra = f"frequency = {num(s['frequency'])}" if s["kind"] == "periodic" else f"count = {num(s['count'])}"
        out += ["[[pedestrians]]", f"origin = {p['origin']}", f"destination = {p['destination']}",
                f'spawn = {{ kind = "{s["kind"]}", {extra} }}']
    DST.write_text("\n".join(out) + "\n")
    back = tomli.loads(DST.read_text())
    assert back == d, "fixture does not parse back to the same data"
    print(f"wrote {DST}: {len(d['obstacles'])} obstacles, {len(d['waypoints'])} waypoints")


if __name__ == "__main__":
    main()
