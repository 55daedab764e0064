"""Why there is no GPU field builder: iterate the reference's local eikonal update
(field.rs:173-186, as restated in oracle/oracle_field.c) to its fixed point in parallel sweeps
and compare with the heap-ordered fast marching the reference runs.  CPU only.
    python tools/fmm_order_dependence.py"""
import sys, numpy as np
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / 'tests'))
from oracle import pyoracle as oracle
import helpers
sc = helpers.random_obstacle_scenario(60.0, 40)
field = helpers.oracle_field(oracle, sc)
ref = field.distance_map.astype(np.float32)
rows, cols = ref.shape
f = np.float32(field.unit)
u = np.where(ref == 0, np.float32(0), np.float32(1e24)).astype(np.float32)
src = ref == 0
BIG = np.float32(3.4028235e38)
def sweep(u):
    p = np.pad(u, 1, constant_values=BIG)
    u1 = np.minimum(p[1:-1, :-2], p[1:-1, 2:])   # x neighbours
    u2 = np.minimum(p[:-2, 1:-1], p[2:, 1:-1])   # y neighbours
    d = u1 - u2
    sq = np.float32(2) * f * f - d * d
    with np.errstate(invalid='ignore', over='ignore'):
        quad = (u1 + u2 + np.sqrt(np.maximum(sq, 0))) / np.float32(2)
    one = np.minimum(u1, u2) + f
    cand = np.where(sq >= 0, quad, one).astype(np.float32)
    new = np.minimum(u, cand)
    new[src] = 0
    return new
for it in range(2000):
    n = sweep(u)
    if np.array_equal(n, u): break
    u = n
print("iterations", it, "cells", u.size)
diff = u != ref
print("cells differing from heap FMM:", int(diff.sum()), "max rel", float(np.max(np.abs(u - ref)[diff] / ref[diff])) if diff.any() else 0.0)
