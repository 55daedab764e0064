#!/usr/bin/env python3
"""bench.py -- agent-steps/s of the per-timestep pedestrian update on MI355X.

Workload (BASELINE.json `metric`: "agent-steps/sec at N=1e6"; configs[2]):
a synthetic uniform crowd of 1e6 agents per GPU in a sparse.toml-style box at 1 agent/m^2
(1000 m x 1000 m per GPU; with G GPUs the box is 1000 m x G*1000 m and is cut into G row
bands -- weak scaling, per-GPU work fixed), default SimulatorOptions (neighbor grid 1.4 m,
distance/potential maps at 0.25 m built by the product's own Field::from_scenario), all
agents heading for the right-hand waypoint.  One "step" = one Simulator::tick of the hot
path: sort/despawn pass (spawn_pedestrians with no new agents) + update_states.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...

Prints ONE JSON line on rank 0.  `value` is whole-job agent-steps/s with the state
resident in HBM; `roofline` prices the dominant kernel (force + integrate) at its
algorithmic 40 B/agent against 8 TB/s; `cpu_baseline` times the CPU oracle (a C port of
the reference's Rust/rayon path -- the Rust binary cannot be built here) on the host cores.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

AGENTS_PER_GPU = 1_000_000
DENSITY = 1.0                      # agents / m^2
BYTES_FORCE = 40                   # 24 B read + 16 B written per agent
BYTES_TICK = 88                    # + sort/reorder pass 24 R + 24 W
HBM_PEAK_GBS = 8000.0              # MI355X_MICROARCH.md: 8.0 TB/s spec


def box_geometry(width: float, height: float, wall_w: float = 0.2, margin: float = 10.0):
    """sparse.toml-style box: 4 thin border walls, waypoint lines `margin` m inside the
    left / right edges.  Rows of (x0, y0, x1, y1, width)."""
    obstacles = np.array([[0, 0, 0, height, wall_w], [width, 0, width, height, wall_w],
                          [0, 0, width, 0, wall_w], [0, height, width, height, wall_w]], np.float32)
    waypoints = np.array([[margin, margin, margin, height - margin, 1.0],
                          [width - margin, margin, width - margin, height - margin, 1.0]], np.float32)
    return obstacles, waypoints


def uniform_crowd(n: int, x_range, y_range, seed: int):
    """Seeded synthetic crowd: positions uniform in the rectangle, destination = right
    waypoint, v0 ~ N(1.34, 0.26) clipped to [0.5, 2.2], velocity = 0.5 * v0 along +x."""
    rng = np.random.default_rng(seed)
    pos = np.empty((n, 2), np.float32)
    pos[:, 0] = rng.uniform(x_range[0], x_range[1], n)
    pos[:, 1] = rng.uniform(y_range[0], y_range[1], n)
    dest = np.ones(n, np.uint32)
    v0 = np.clip(rng.normal(1.34, 0.26, n), 0.5, 2.2).astype(np.float32)
    vel = np.zeros((n, 2), np.float32)
    vel[:, 0] = 0.5 * v0
    return pos, dest, v0, vel


def free_space_crowd(field, size, n, n_dest, seed, dest_rule=None):
    """Seeded crowd in free space (distance map > 0.6 m): v0 ~ N(1.34, 0.26) clipped, velocity
    0.5 * v0 towards +x / -x by destination parity."""
    rng = np.random.default_rng(seed)
    dm = field.distance_map
    pos = np.zeros((0, 2), np.float32)
    while len(pos) < n:
        p = rng.uniform([2.0, 2.0], [size[0] - 2.0, size[1] - 2.0], (int((n - len(pos)) * 1.3) + 1000, 2)
                        ).astype(np.float32)
        iy, ix = (p[:, 1] / field.unit).astype(int), (p[:, 0] / field.unit).astype(int)
        pos = np.concatenate([pos, p[dm[iy, ix] > 0.6]])[:n]
    dest = dest_rule(pos) if dest_rule else rng.integers(0, n_dest, n).astype(np.uint32)
    v0 = np.clip(rng.normal(1.34, 0.26, n), 0.5, 2.2).astype(np.float32)
    vel = np.zeros((n, 2), np.float32)
    vel[:, 0] = np.where(dest % 2 == 1, 0.5, -0.5) * v0
    return pos, dest.astype(np.uint32), v0, vel


def other_workload(name):
    """BASELINE.json configs[1] (C2) and configs[3] (C4) as optional bench workloads."""
    if name == "c2":       # scenarios/random.toml (data fixture): 200 x 200 m, 4 waypoints, 1004 walls
        from pedoni_amd import scenario as scn
        sc = scn.load(ROOT / "tests" / "golden" / "scenarios" / "random.toml")
        L = float(sc.field.size[0])
        obstacles, waypoints = sc.obstacle_array(), sc.waypoint_array()
        crowd = lambda field: free_space_crowd(field, (L, L), 100_000, 4, seed=100)
        return obstacles, waypoints, (L, L), crowd, \
            "scenarios/random.toml geometry (200x200 m, 1004 obstacles), N=1e5 injected agents (rho~2.5/m^2), 4 destinations"
    # bottleneck.toml geometry x5 (tests/golden/scenarios/bottleneck_x5.toml)
    obstacles = np.array([[250, 0, 500, 450, 25], [250, 1000, 500, 550, 25], [750, 0, 500, 450, 25],
                          [750, 1000, 500, 550, 25]], np.float32)
    waypoints = np.array([[50, 50, 50, 950, 1], [950, 50, 950, 950, 1]], np.float32)
    rule = lambda pos: (pos[:, 0] < 500.0).astype(np.uint32)   # counter-flow halves
    crowd = lambda field: free_space_crowd(field, (1000.0, 1000.0), 1_000_000, 2, seed=4, dest_rule=rule)
    path = "explicit wall segments" if name == "c4seg" else "distance map"
    return obstacles, waypoints, (1000.0, 1000.0), crowd, \
        f"bottleneck x5 (1000x1000 m, 4 funnel walls), N=1e6 counter-flow, obstacle force via {path}"


def cpu_baseline(size, field, obstacles, pos, dest, v0, vel, budget_s: float = 20.0,
                 use_distance_map: bool = True):
    """Time the CPU oracle (oracle/, C port of the reference's CPU path with its parallel
    structure: serial sort/despawn, parallel-for accelerations, serial integrator) on a
    bounded sample of the SAME workload.  Checker code, used here only as the baseline."""
    from oracle import pyoracle
    cores = os.cpu_count() or 1
    try:
        cores = len(os.sched_getaffinity(0))
    except Exception:
        pass
    cores = min(cores, 32)   # a 1-GPU box's CPU share; the serial passes dominate anyway
    ofield = pyoracle.Field(field.unit, field.distance_map, field.potential_maps)
    m = pyoracle.OracleModel(size, threads=cores, use_distance_map=use_distance_map)
    m.spawn_pedestrians(ofield, pos, dest, v0, vel)
    steps, t_used = 0, 0.0
    t0 = time.perf_counter()
    m.update_states(ofield, obstacles)       # first step (also warms caches)
    m.spawn_pedestrians(ofield)
    per = time.perf_counter() - t0
    n_steps = int(max(2, min(400, budget_s / max(per, 1e-3))))
    t0 = time.perf_counter()
    agents = 0
    for _ in range(n_steps):
        agents += m.get_pedestrian_count()
        m.update_states(ofield, obstacles)
        m.spawn_pedestrians(ofield)
        steps += 1
    t_used = time.perf_counter() - t0
    return {
        "value": agents / t_used, "unit": "agent-steps/s", "cores": cores, "kind": "port",
        "sample": f"{steps} ticks of the same {len(pos)}-agent crowd and field "
                  f"({t_used:.1f} s; C port of pedoni's SocialForceModel CPU path with OpenMP where "
                  "upstream uses rayon -- the Rust binary cannot be built here)",
    }


def _by_age(path):
    """Sort key of profiles/rNN_vM_* names: numeric fields compare as numbers (v11 after v9)."""
    import re
    return [int(t) if t.isdigit() else t for t in re.split(r"(\d+)", path.name)]


def pmc_traffic(workload: str):
    """HBM bytes per force-kernel launch from the newest committed rocprofv3 --pmc passes
    (profiles/*pmc_force*.json written by tools/pmc_summary.py) for this workload, with the
    profile it came from -- (bytes, tag) or (None, None)."""
    best = (None, None)
    for p in sorted((ROOT / "profiles").glob("*pmc_force*.json"), key=_by_age):      # rNN_vM names, oldest first
        try:
            d = json.loads(p.read_text())
        except Exception:
            continue
        if d.get("workload") == workload and d.get("kernel", "").startswith("force"):
            best = (d.get("hbm_bytes_per_launch"), p.name)
    return best


# VALU issue on gfx950 (tools/microbench/valu_issue.hip, profiles/r02_valu_issue.txt): a SIMD-32
# issues one wave64 VALU instruction per 2 cycles at best (MI355X_MICROARCH.md); with the 6
# waves per SIMD the force kernel runs at, independent v_fma_f32 streams measured 2.3-2.4.
VALU_CYCLES_PER_INST = 2.0
N_SIMDS = 1024


def valu_floor(avg_launch_ms: float):
    """Instruction-issue floor of the force kernel from the newest committed stall-counter
    profile (profiles/*_stalls.json, tools/profile_stalls.sh): SQ_INSTS_VALU wave-instructions
    per launch x 2 cycles / 1024 SIMDs, priced at the clock the profiled launches held
    (GRBM_GUI_ACTIVE / 8 XCDs per launch / its SQ_BUSY time is not available, so the clock is
    cycles per launch / the profiled launch duration)."""
    best = None
    names = sorted((ROOT / "profiles").glob("*_stalls.json"),
                   key=lambda q: (0 if "_base_" in q.name else 1, _by_age(q)))             # rNN_vM names, oldest first
    for p in names:
        try:
            d = json.loads(p.read_text())
        except Exception:
            continue
        for k, v in d.items():
            if "force_kernel_queue" in k and "<0," in k and v.get("SQ_INSTS_VALU") and v.get("GRBM_GUI_ACTIVE"):
                best = (v, p.name)
    if not best:
        return None
    v, tag = best
    insts = v["SQ_INSTS_VALU"]
    floor_cycles = insts * VALU_CYCLES_PER_INST / N_SIMDS
    launch_cycles = v["GRBM_GUI_ACTIVE"] / 8.0           # rocprofv3 sums the 8 XCDs
    clock_ghz = v.get("clock_ghz") or 2.1
    floor_ms = floor_cycles / (clock_ghz * 1e6)
    return {"insts_per_launch": insts, "cycles_per_inst": VALU_CYCLES_PER_INST, "simds": N_SIMDS,
            "floor_ms": floor_ms, "frac": floor_ms / avg_launch_ms,
            "frac_in_profiled_run": floor_cycles / launch_cycles, "clock_ghz": clock_ghz, "profile": tag}


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--agents-per-gpu", type=int, default=AGENTS_PER_GPU)
    ap.add_argument("--math", choices=["exact", "fast"], default="exact")
    ap.add_argument("--workload", choices=["c3", "c2", "c4", "c4seg"], default="c3",
                    help="c3 = BASELINE metric workload (default, the only one the driver runs); "
                         "c2 = random-obstacle field with 1e5 agents; c4 = bottleneck x5 with 1e6 "
                         "agents (distance map); c4seg = same with explicit wall segments")
    ap.add_argument("--density", type=float, default=DENSITY,
                    help="c3 only: agents per m^2 (SURVEY 8(d) sweeps 0.25 / 1 / 4; the metric is quoted at 1)")
    ap.add_argument("--work-size", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-fast-leg", action="store_true",
                    help="skip the secondary PEDONI_MATH_FAST measurement of the same crowd")
    ap.add_argument("--cpu-budget", type=float, default=20.0)
    ap.add_argument("--no-profile", action="store_true",
                    help="skip the per-kernel hipEvent pairs in the timed region")
    args = ap.parse_args()

    # the host driver of this pool only supports dmabuf IPC (RCCL / cross-process tensors)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    # RCCL prints a version banner on stdout: keep fd 1 for the ONE JSON line
    json_fd = os.dup(1)
    os.dup2(2, 1)
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus N>1 must be launched with torch.distributed.run "
                     "(one rank per GPU)")
        args.gpus = world

    import torch
    from pedoni_amd import abi, host

    if not torch.cuda.is_available() or abi.device_count() < 1:
        sys.exit("bench.py needs a HIP device: the backend has no CPU fallback")
    local_rank %= max(torch.cuda.device_count(), 1)   # (a rehearsal may share one card)
    torch.cuda.set_device(local_rank)
    # PEDONI_FORCE_SHARDED=1: run the row-band path even with one rank (smoke test of the
    # RCCL / stream plumbing on a single GPU); never set by the driver
    force_sharded = os.environ.get("PEDONI_FORCE_SHARDED") == "1"
    dist = None
    if world > 1 or force_sharded:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        import torch.distributed as dist
        # "nccl" IS RCCL on ROCm; PEDONI_DIST_BACKEND=gloo only rehearses the code path
        backend = os.environ.get("PEDONI_DIST_BACKEND", "nccl")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)
    stream = torch.cuda.Stream()
    torch.cuda.set_stream(stream)   # kernels + collective on one non-default stream

    G = args.gpus
    n_per = args.agents_per_gpu
    density = args.density
    side = float(np.sqrt(n_per / density))          # 1000 m at 1e6 agents and rho = 1
    width, height = side, side * G
    obstacles, waypoints = box_geometry(width, height)
    workload = (f"uniform crowd N={n_per * G:.0e} ({n_per:.0e}/GPU) in a {width:.0f}x{height:.0f} m box, "
                f"rho={density:g}/m^2, neighbor grid 1.4 m, field maps 0.25 m, fp32").replace("e+0", "e")

    custom_crowd = None
    if args.workload != "c3":
        if G != 1:
            sys.exit("--workload c2/c4 are single-GPU configurations")
        obstacles, waypoints, (width, height), custom_crowd, workload = other_workload(args.workload)
        n_per = 100_000 if args.workload == "c2" else 1_000_000

    t0 = time.perf_counter()
    field = host.Field.build((width, height), 0.25, obstacles, waypoints)
    t_field = time.perf_counter() - t0

    opt = abi.Options(math_mode=abi.MATH_FAST if args.math == "fast" else abi.MATH_EXACT,
                      gpu_work_size=args.work_size, initial_capacity=int(n_per * 1.3),
                      use_distance_map=args.workload != "c4seg")
    map_rows = bounds = None
    if G > 1 or force_sharded:
        from pedoni_amd.sharded import ShardedModel, band_rows, default_halo_cap
        rows = int(np.ceil(np.float32(height) / np.float32(opt.neighbor_grid_unit)))   # neighbor_grid.rs:14-20
        bounds = band_rows(rows, G)                # uniform crowd: equal rows = equal agents
        # each rank uploads only its band's texel rows of the three maps (1/G of 0.5 GB each at G = 8)
        map_rows = abi.shard_map_rows(bounds[rank], bounds[rank + 1], 0, opt.neighbor_grid_unit, field.unit,
                                      field.shape[0])
    model = abi.HipModel(opt, (width, height), field.distance_map, field.potential_maps,
                         field.unit, obstacles, device=local_rank, map_rows=map_rows)

    exchange = None
    if G > 1 or force_sharded:
        model.set_stream(stream.cuda_stream)
        assert model.neighbor_grid_shape()[0] == rows
        cap = default_halo_cap(int(width * 1.4 * density))
        runner = shard = None
        if os.environ.get("PEDONI_EXCHANGE", "rccl") == "rccl" and dist.get_backend() == "nccl":
            # the driver below the C-ABI: libpedoni_hip owns an RCCL communicator and sends /
            # receives the lists itself (ncclSend / ncclRecv with rank +- 1 on the model's
            # stream).  torch.distributed only carries the 128-byte id and the timing barrier.
            # (every rank takes part in every collective below whatever failed locally)
            def agreed(ok: int) -> bool:
                flag = torch.tensor([ok], dtype=torch.int32, device="cuda")
                dist.all_reduce(flag, op=dist.ReduceOp.MIN)
                return int(flag.item()) == 1

            ok, why = 1, ""
            idt = torch.zeros(abi.SHARD_ID_BYTES, dtype=torch.uint8, device="cuda")
            if rank == 0:
                try:
                    idt.copy_(torch.frombuffer(bytearray(abi.shard_unique_id()), dtype=torch.uint8))
                except Exception as e:             # noqa: BLE001 -- an all-zero id tells the others
                    ok, why = 0, str(e)
            dist.broadcast(idt, 0)
            uid = bytes(idt.cpu().numpy().tobytes())
            if not any(uid):
                ok, why = 0, why or "rank 0 could not create an RCCL id"
            if agreed(ok):
                try:
                    shard = abi.Shard(model, rank, G, bounds, cap, unique_id=uid)   # ncclCommInitRank
                except Exception as e:             # noqa: BLE001
                    ok, why = 0, str(e)
                if agreed(ok):
                    try:
                        shard.selftest()           # a token ring through the very send / recv pair
                    except Exception as e:         # noqa: BLE001
                        ok, why = 0, str(e)
                    ok = 1 if agreed(ok) else 0
                else:
                    ok = 0
            else:
                ok = 0
            if ok == 1:
                exchange = "direct RCCL ncclSend/ncclRecv to rank+-1, driven by libpedoni_hip (pedoni_shard_tick_n)"
            else:
                print(f"[bench] rank {rank}: direct RCCL path unavailable ({why or 'another rank failed'}); "
                      "falling back to torch.distributed all_gather", file=sys.stderr)
                if shard is not None:
                    shard.close()
                shard = None
                model.close()
                model = abi.HipModel(opt, (width, height), field.distance_map, field.potential_maps,
                                     field.unit, obstacles, device=local_rank, map_rows=map_rows)
                model.set_stream(stream.cuda_stream)
        if shard is None:
            runner = ShardedModel(model, rank, G, dist, torch, halo_cap=cap, bounds=bounds,
                                  overlap=os.environ.get("PEDONI_OVERLAP") == "1")
            exchange = "torch.distributed all_gather_into_tensor (RCCL), driven from Python"
        lo, hi = bounds[rank], bounds[rank + 1]
        # this rank's agents: exactly its own band of grid rows (2 m clear of the outer walls)
        y_lo, y_hi = lo * 1.4 + 0.01, hi * 1.4 - 0.01
    else:
        runner = shard = None
        y_lo, y_hi = 0.0, height
    if custom_crowd is not None:
        pos, dest, v0, vel = custom_crowd(field)
    else:
        pos, dest, v0, vel = uniform_crowd(
            n_per, (12.0, width - 12.0), (max(y_lo, 2.0), min(y_hi, height - 2.0)), seed=12345 + rank)

    if shard is not None:
        model.append(pos, dest, v0, vel)
        shard.begin()
        step_fn = shard.tick_n
        # plain tick (exchange, then the whole update) or overlapped (the next exchange under the
        # interior rows' update)?  Which is faster depends on what the exchange costs on this
        # node: time 20 ticks of each, every rank keeps the mode that was faster for the slowest.
        mode_ms = {}
        for mode in (False, True):
            shard.set_overlap(mode)
            shard.tick_n(5)
            dist.barrier(); torch.cuda.synchronize()
            t0 = time.perf_counter()
            shard.tick_n(20)
            torch.cuda.synchronize()
            t = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device="cuda")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            mode_ms[mode] = float(t.item()) / 20 * 1e3
        use_overlap = mode_ms[True] < mode_ms[False]
        shard.set_overlap(use_overlap)
        exchange += (f"; tick form: {'overlapped' if use_overlap else 'plain'} "
                     f"(probe: plain {mode_ms[False] * 1e3:.0f} us, overlapped {mode_ms[True] * 1e3:.0f} us per tick; "
                     "its 50 ticks precede the warmup, so the timed crowd is 50 ticks older than a 1-GPU run's)")
    elif runner is not None:
        assert (runner.owner_of(pos[:, 1]) == rank).all()
        runner.load(pos, dest, v0, vel)
        step_fn = runner.tick_n
    else:
        model.append(pos, dest, v0, vel)
        step_fn = model.tick_n

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    step_fn(args.warmup)
    barrier()
    sharded = runner is not None or shard is not None
    n_before = model.owned_count() if sharded else model.get_pedestrian_count()
    if not args.no_profile:
        # inside the timed region the dominant kernel is event-timed on every 9th tick only (an odd period: the 8 ticks between two timed ones replay as 4 captured pairs):
        # timed ticks launch eagerly (+ one hipEvent pair), the others replay the captured tick
        # pair; the full per-kernel breakdown is a separate pass after the timed region
        model.profile(True, kernels=[abi.K_FORCE], every=9)
        model.kernel_times(reset=True)
    barrier()
    t0 = time.perf_counter()
    step_fn(args.steps)
    barrier()
    elapsed = time.perf_counter() - t0
    ktimes = model.kernel_times(reset=True) if not args.no_profile else {}
    model.profile(False)
    n_after = model.owned_count() if sharded else model.get_pedestrian_count()
    breakdown = {}
    if not args.no_profile:
        model.profile(True)
        step_fn(min(args.steps, 20))
        barrier()
        bt = model.kernel_times(reset=True)
        model.profile(False)
        breakdown = {k: v["total_ms"] / max(v["launches"], 1) * (v["launches"] / min(args.steps, 20))
                     for k, v in bt.items() if v["launches"]}

    agents_local = 0.5 * (n_before + n_after)       # despawns during the run are negligible
    if dist is not None and world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        a = torch.tensor([agents_local], dtype=torch.float64, device="cuda")
        dist.all_reduce(a, op=dist.ReduceOp.SUM)
        agents_total = float(a.item())
    else:
        agents_total = agents_local

    if rank == 0:
        value = agents_total * args.steps / elapsed
        out = {
            "metric": "agent-steps/sec at N=1e6; achieved HBM GB/s vs roofline; 1/2/4/8-GPU scaling",
            "value": value, "unit": "agent-steps/s", "n_gpus": G, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": workload, "agents_total": int(round(agents_total)),
                       "math_mode": args.math, "parallelism": f"row-bands x{G}; {exchange}" if sharded else "single GPU",
                       "field_build_s": round(t_field, 2),
                       "tick_algorithmic_GBps": BYTES_TICK * value / 1e9},
        }
        fk = ktimes.get("force_integrate")
        if fk and fk["launches"]:
            avg_ms = fk["total_ms"] / fk["launches"]
            achieved = BYTES_FORCE * agents_local / (avg_ms * 1e-3) / 1e9
            traffic, traffic_tag = pmc_traffic(workload)
            valu = valu_floor(avg_ms) if args.math == "exact" and args.workload == "c3" and G == 1 else None
            hbm_frac = achieved / HBM_PEAK_GBS
            out["roofline"] = {
                # the roof the kernel is closer to: HBM bytes at 8 TB/s, or VALU issue at one
                # wave instruction per 2 cycles per SIMD (no MFMA on this path)
                "bound": "valu" if valu and valu["frac"] > hbm_frac else "hbm",
                "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": hbm_frac, "traffic": traffic, "traffic_profile": traffic_tag,
                "kernel": "force_integrate", "avg_launch_ms": avg_ms,
                "timed_launches": fk["launches"],
                "algorithmic_bytes_per_launch": BYTES_FORCE * agents_local,
                "valu": valu,
            }
            out["kernel_ms_per_step"] = breakdown  # separate pass, every kernel event-timed
        else:
            out["roofline"] = None
        if G == 1 and args.math == "exact" and not args.no_fast_leg:
            # the same crowd in PEDONI_MATH_FAST (hardware rcp / rsq / sqrt / exp with exact
            # decisions): inside north_star's 1e-5 per step at these sizes
            # (tests/test_gpu_fullsize.py::test_fast_math_*), reported beside the headline,
            # which stays the bit-identical exact mode
            fopt = abi.Options(math_mode=abi.MATH_FAST, gpu_work_size=args.work_size,
                               initial_capacity=int(n_per * 1.3), use_distance_map=args.workload != "c4seg")
            fm = abi.HipModel(fopt, (width, height), field.distance_map, field.potential_maps, field.unit,
                              obstacles, device=local_rank)
            fm.append(pos, dest, v0, vel)
            fm.tick_n(args.warmup)
            fm.synchronize()
            nb = fm.get_pedestrian_count()
            fm.profile(True, kernels=[abi.K_FORCE], every=9)
            fm.kernel_times(reset=True)
            t0 = time.perf_counter()
            fm.tick_n(args.steps)
            fm.synchronize()
            el = time.perf_counter() - t0
            fk2 = fm.kernel_times(reset=True).get("force_integrate", {})
            fm.profile(False)
            na = fm.get_pedestrian_count()
            out["fast_math"] = {
                "value": 0.5 * (nb + na) * args.steps / el, "unit": "agent-steps/s",
                "ms_per_step": 1e3 * el / args.steps, "math_mode": "fast",
                "force_avg_launch_ms": fk2["total_ms"] / fk2["launches"] if fk2.get("launches") else None,
                "tolerance": "1e-5 relative per step from identical state (north_star); exact mode is bit-identical",
            }
            fm.close()
        if G == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline((width, height), field, obstacles, pos, dest, v0, vel,
                                               args.cpu_budget, args.workload != "c4seg")
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(out) + "\n").encode())

    if shard is not None:
        shard.close()                              # ncclCommDestroy before the model goes
    model.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
