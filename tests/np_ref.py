"""Independent NumPy-float32 restatement of the social-force tick, written from the
formulas of SURVEY.md section 8(a) (Helbing-Molnar elliptical specification), NOT from
the C oracle's code: a second opinion that must agree with oracle/ bit for bit.

Scalar np.float32 arithmetic rounds every operation to fp32 and never fuses a*b+c.
exp goes through the host libm (what Rust's f32::exp calls).  Small N only.
"""
import ctypes
import ctypes.util

import numpy as np

F = np.float32
_libm = ctypes.CDLL(ctypes.util.find_library("m") or "libm.so.6")
_libm.expf.restype = ctypes.c_float
_libm.expf.argtypes = [ctypes.c_float]

COS_PHI = F(-0.17364817766693036)
V0_OVER_SIGMA = F(2.1) / F(0.3)      # 6.9999995, not 7.0
U0R = F(10.0) * F(0.2)               # exactly 2.0


def expf(x):
    return F(_libm.expf(float(x)))


def as_i32(v):
    """Rust `f32 as i32`."""
    v = float(v)
    if v != v:
        return 0
    return int(max(min(np.trunc(v), 2147483647), -2147483648))


def bilinear(grid, px, py):
    px, py = F(px), F(py)
    with np.errstate(invalid="ignore"):
        bx, by = np.floor(px), np.floor(py)
        tx, ty = px - bx, py - by
        sx, sy = F(1) - tx, F(1) - ty
    ix, iy = as_i32(bx), as_i32(by)

    def tex(x, y):
        if x < 0 or y < 0 or y >= grid.shape[0] or x >= grid.shape[1]:
            return F(1e12)
        return F(grid[y, x])

    with np.errstate(invalid="ignore", over="ignore"):
        acc = F(0)
        acc = acc + sy * sx * tex(ix, iy)
        acc = acc + sy * tx * tex(ix + 1, iy)
        acc = acc + ty * sx * tex(ix, iy + 1)
        acc = acc + ty * tx * tex(ix + 1, iy + 1)
    return acc


def sobel(grid, px, py):
    px, py = F(px), F(py)
    m1, p1, z = F(-1), F(1), F(0)
    u = {(r, c): bilinear(grid, px + dc, py + dr)
         for r, dr in ((0, m1), (1, z), (2, p1)) for c, dc in ((0, m1), (1, z), (2, p1))
         if (r, c) != (1, 1)}
    with np.errstate(invalid="ignore", over="ignore"):
        gx = u[0, 0] + u[1, 0] + u[1, 0] + u[2, 0] - u[0, 2] - u[1, 2] - u[1, 2] - u[2, 2]
        gy = u[0, 0] + u[0, 1] + u[0, 1] + u[0, 2] - u[2, 0] - u[2, 1] - u[2, 1] - u[2, 2]
    return gx, gy


def field_coord(unit, x, y):
    return F(x) / F(unit) - F(0.5), F(y) / F(unit) - F(0.5)


def normalize(x, y):
    with np.errstate(all="ignore"):
        r = F(1) / np.sqrt(x * x + y * y)
        return x * r, y * r


def sort_despawn(pos, dest, vel, v0, field_unit, potential_maps, grid_unit, grid_shape):
    """Stable sort by cell id of the agents that are inside the grid and whose potential
    exceeds 0.25; returns sorted arrays and the per-cell prefix counts."""
    rows, cols = grid_shape
    keys, keep = [], []
    for i in range(len(pos)):
        with np.errstate(all="ignore"):
            cx, cy = as_i32(F(pos[i, 0]) / F(grid_unit)), as_i32(F(pos[i, 1]) / F(grid_unit))
        if cx < 0 or cy < 0 or cy >= rows or cx >= cols:
            continue
        q = field_coord(field_unit, pos[i, 0], pos[i, 1])
        if not (bilinear(potential_maps[dest[i]], *q) > F(0.25)):
            continue
        keys.append(cy * cols + cx)
        keep.append(i)
    keys, keep = np.array(keys, np.int64), np.array(keep, np.int64)
    order = np.argsort(keys, kind="stable")
    idx = keep[order] if len(keep) else keep
    counts = np.bincount(keys, minlength=rows * cols) if len(keys) else np.zeros(rows * cols, np.int64)
    cell_start = np.concatenate([[0], np.cumsum(counts)]).astype(np.uint32)
    return pos[idx], dest[idx], vel[idx], v0[idx], cell_start


def pair_force(px, py, ex, ey, qx, qy, wx, wy):
    """Force on the agent at p (goal direction e) from a neighbour at q moving with w."""
    with np.errstate(all="ignore"):
        dx, dy = px - qx, py - qy
        d2 = dx * dx + dy * dy
        if d2 > F(4.0):
            return None
        d = np.sqrt(d2)
        rinv = F(1) / d
        nx, ny = dx * rinv, dy * rinv
        t1x, t1y = dx - wx * F(0.1), dy - wy * F(0.1)
        l1 = np.sqrt(t1x * t1x + t1y * t1y)
        t2 = d + l1
        step = np.sqrt(wx * wx + wy * wy) * F(0.1)
        b = np.sqrt(t2 * t2 - step * step) * F(0.5)
        gx = (t2 * (nx + t1x / l1)) / (F(4.0) * b)
        gy = (t2 * (ny + t1y / l1)) / (F(4.0) * b)
        k = V0_OVER_SIGMA * expf(-b / F(0.3))
        fx, fy = k * gx, k * gy
        if ex * (-fx) + ey * (-fy) < np.sqrt(fx * fx + fy * fy) * COS_PHI:
            fx, fy = fx * F(0.5), fy * F(0.5)
    return fx, fy


def accelerations(pos, dest, vel, v0, cell_start, field_unit, distance_map, potential_maps,
                  grid_unit, grid_shape):
    rows, cols = grid_shape
    n = len(pos)
    acc = np.zeros((n, 2), np.float32)
    for a in range(n):
        px, py = F(pos[a, 0]), F(pos[a, 1])
        q = field_coord(field_unit, px, py)
        with np.errstate(all="ignore"):
            ex, ey = normalize(*sobel(potential_maps[dest[a]], *q))
            ax = (ex * F(v0[a]) - F(vel[a, 0])) / F(0.5)
            ay = (ey * F(v0[a]) - F(vel[a, 1])) / F(0.5)
            cx, cy = as_i32(px / F(grid_unit)), as_i32(py / F(grid_unit))
            for y in range(max(cy - 1, 0), min(cy + 1, rows - 1) + 1):
                lo = cell_start[y * cols + max(cx - 1, 0)]
                hi = cell_start[y * cols + min(cx + 1, cols - 1) + 1]
                for i in range(lo, hi):
                    if i == a:
                        continue
                    f = pair_force(px, py, ex, ey, F(pos[i, 0]), F(pos[i, 1]), F(vel[i, 0]), F(vel[i, 1]))
                    if f is not None:
                        ax, ay = ax + f[0], ay + f[1]
            dist = bilinear(distance_map, *q)
            gx, gy = normalize(*sobel(distance_map, *q))
            k = U0R * expf(-dist / F(0.2))
            ax, ay = ax + k * (-gx), ay + k * (-gy)
        acc[a] = (ax, ay)
    return acc


def integrate(pos, vel, v0, acc):
    pos, vel = pos.copy(), vel.copy()
    with np.errstate(all="ignore"):
        for i in range(len(pos)):
            pvx, pvy = F(vel[i, 0]), F(vel[i, 1])
            vx, vy = pvx + F(acc[i, 0]) * F(0.1), pvy + F(acc[i, 1]) * F(0.1)
            m = F(v0[i]) * F(1.3)
            l2 = vx * vx + vy * vy
            if l2 > m * m:
                l = np.sqrt(l2)
                vx, vy = m * (vx / l), m * (vy / l)
            vel[i] = (vx, vy)
            pos[i] = (F(pos[i, 0]) + (vx + pvx) * F(0.05), F(pos[i, 1]) + (vy + pvy) * F(0.05))
    return pos, vel
