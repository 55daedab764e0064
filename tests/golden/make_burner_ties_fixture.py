#!/usr/bin/env python3
"""Writes tests/golden/burner_corner_ties.json: for every scenario file (the data fixtures under
tests/golden/scenarios and, in the build container, the reference's scenarios/*.toml -- inputs
only), the cells of the obstacle mask at 0.25 m where this build's line burner (GDAL all-touched
lineage, what geo-rasterize 0.1.2 documents) and an exact grid traversal of the same outlines
DISAGREE.  Every one is a corner / end-point tie.  The rasteriser is PARITY UNPINNED
(field.rs:42-88; upstream's test of it only prints): these cells are exactly where a future pin
of geo-rasterize -- its own output on these scenarios -- would have to be looked at first.

    python tests/golden/make_burner_ties_fixture.py          (from the repo root)
"""
import json
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))

from pedoni_amd import host, scenario as scn      # noqa: E402
import test_host_cpu as t                          # noqa: E402  (_exact_traversal, _outline_cells)


def ties(path):
    sc = scn.load(path)
    got = host.Field.build(sc.field.size, 0.25, sc.obstacle_array(), sc.waypoint_array())
    rows, cols = got.shape
    exact = np.zeros((rows, cols), bool)
    exact[0, :] = exact[-1, :] = exact[:, 0] = exact[:, -1] = True
    for seg in sc.obstacle_array():
        exact |= t._exact_traversal(t._outline_cells(seg), rows, cols)
    rr, cc = np.nonzero(exact != got.obstacle_exist)
    # the waypoint outlines (field.rs:66-88): the zero set of each potential map
    wp = []
    for k, seg in enumerate(sc.waypoint_array()):
        zero = got.potential_maps[k] == 0
        ex = t._exact_traversal(t._outline_cells(seg), rows, cols)
        wr, wc = np.nonzero(ex != zero)
        wp += [[int(k), int(r), int(c), int(zero[r, c])] for r, c in zip(wr, wc)]
    return {"shape": [rows, cols], "burnt_cells": int(got.obstacle_exist.sum()),
            # [row, col, this build burns it (1) / only the exact traversal does (0)]
            "cells": [[int(r), int(c), int(got.obstacle_exist[r, c])] for r, c in zip(rr, cc)],
            # [waypoint, row, col, this build burns it (1) / only the exact traversal does (0)]
            "waypoint_cells": wp}


def main():
    files = {p.name: p for p in sorted((ROOT / "tests" / "golden" / "scenarios").glob("*.toml"))}
    ref = Path("/root/reference/scenarios")
    if ref.is_dir():
        for p in sorted(ref.glob("*.toml")):
            files.setdefault("reference:" + p.name, p)
    out = {name: ties(p) for name, p in files.items()}
    (ROOT / "tests" / "golden" / "burner_corner_ties.json").write_text(json.dumps(out, indent=1) + "\n")
    print({k: (len(v["cells"]), len(v["waypoint_cells"])) for k, v in out.items()})
    print("obstacle-mask cells", sum(len(v["cells"]) for v in out.values()), "waypoint-outline cells",
          sum(len(v["waypoint_cells"]) for v in out.values()))


if __name__ == "__main__":
    main()
