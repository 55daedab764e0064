#!/usr/bin/env python3
"""bench.py -- agent-steps/s of the per-timestep pedestrian update on MI355X.

Workload (BASELINE.json `metric`: "agent-steps/sec at N=1e6"; configs[2]):
a synthetic uniform crowd of 1e6 agents per GPU in a sparse.toml-style box at 1 agent/m^2
(1000 m x 1000 m per GPU; with G GPUs the box is 1000 m x G*1000 m and is cut into G row
bands -- weak scaling, per-GPU work fixed), default SimulatorOptions (neighbor grid 1.4 m,
distance/potential maps at 0.25 m built by the product's own Field::from_scenario), all
agents heading for the right-hand waypoint.  One "step" = one Simulator::tick of the hot
path: sort/despawn pass (spawn_pedestrians with no new agents) + update_states.

    python bench.py --gpus N --steps K --warmup W          (N > 1: starts its own N ranks as a child)
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
SETTLE_TICKS = 58                  # untimed ticks every run has done before its warmup (see main)


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
    host_cores = cores
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
        # `cores` = the OpenMP threads the parallel-for really ran on; the box offers this process:
        "host_cores_available": host_cores, "host_cores_total": os.cpu_count(),
        "sample": f"{steps} ticks of the same {len(pos)}-agent crowd and field on {cores} of the box's "
                  f"{host_cores} usable host cores "
                  f"({t_used:.1f} s; C port of pedoni's SocialForceModel CPU path with OpenMP where "
                  "upstream uses rayon -- the Rust binary cannot be built here)",
    }


class Watchdog:
    """N > 1 runs only.  A hung collective (a rank that died, a mismatched send / receive) would
    otherwise sit until the driver's own limit and leave no line at all: every stage of the run has
    a bound, and a rank that outlives it says where it was and exits -- a fresh exit(3), never a
    re-exec -- which makes torch.distributed.run take the other ranks down too."""

    def __init__(self, rank: int):
        import threading
        self.rank, self.name, self.deadline = rank, "start", None
        self._t = threading.Thread(target=self._run, daemon=True)
        self._t.start()

    def stage(self, name: str, bound_s: float) -> None:
        self.name, self.deadline = name, time.monotonic() + bound_s

    def done(self) -> None:
        self.deadline = None

    def _run(self) -> None:
        while True:
            time.sleep(0.5)
            d = self.deadline
            if d is not None and time.monotonic() > d:
                print(f"[bench] WATCHDOG: rank {self.rank} exceeded its bound in stage '{self.name}'; "
                      "exiting 3 (no result line)", file=sys.stderr, flush=True)
                os._exit(3)


class SharedField:
    """What bench needs of host.Field, backed by read-only memory maps of /dev/shm files: rank 0
    builds the field ONCE (the 1000 x 8000 m field of an 8-GPU run is 1.5 GB and ~8 s of
    fast marching; 8 ranks building it at once was 12 GB and 8x the threads) and every rank maps
    it; a rank only ever touches the texel rows of its own band."""

    def __init__(self, unit, distance_map, potential_maps):
        self.unit, self.distance_map, self.potential_maps = unit, distance_map, potential_maps
        self.shape = distance_map.shape


def shared_field(dist, torch, ctl, rank, size, unit, obstacles, waypoints):
    """(field, seconds, how): built by rank 0 and mapped by all, or -- if /dev/shm cannot hold it --
    built by every rank as before.  Collective: every rank calls it."""
    from pedoni_amd import host
    tag = f"pedoni_bench_{os.environ.get('MASTER_PORT', '0')}_{os.getuid()}"
    base = Path("/dev/shm") / tag
    t0 = time.perf_counter()
    ok = torch.ones(1, dtype=torch.int32, device=ctl)
    n_maps = torch.zeros(1, dtype=torch.int32, device=ctl)
    if rank == 0:
        try:
            f = host.Field.build(size, unit, obstacles, waypoints)
            base.mkdir(parents=True, exist_ok=True)
            np.save(base / "dm.npy", f.distance_map)
            for k, pm in enumerate(f.potential_maps):
                np.save(base / f"pm{k}.npy", pm)
            n_maps[0] = len(f.potential_maps)
            del f
        except Exception as e:                          # noqa: BLE001 -- e.g. /dev/shm too small
            print(f"[bench] rank 0 could not share the field through /dev/shm ({e}); every rank builds its own",
                  file=sys.stderr)
            ok[0] = 0
    dist.broadcast(ok, 0)
    dist.broadcast(n_maps, 0)
    if int(ok.item()) == 0:
        if rank == 0:
            import shutil
            shutil.rmtree(base, ignore_errors=True)
        return host.Field.build(size, unit, obstacles, waypoints), time.perf_counter() - t0, "built by every rank", None
    dm = np.load(base / "dm.npy", mmap_mode="r")
    pms = [np.load(base / f"pm{k}.npy", mmap_mode="r") for k in range(int(n_maps.item()))]
    return SharedField(unit, dm, pms), time.perf_counter() - t0, "built once by rank 0, mapped from /dev/shm", base


WORKLOAD_KEYS = ("c2", "c4seg", "c4")


def _profile_workload(name: str) -> str:
    """Which bench workload a committed profile was taken on, from its name (rNN_c4_* ...; C3 otherwise)."""
    for k in WORKLOAD_KEYS:
        if f"_{k}_" in name:
            return k
    return "c3"


def _by_age(path):
    """Sort key of profiles/rNN_vM_* names: numeric fields compare as numbers (v11 after v9)."""
    import re
    return [int(t) if t.isdigit() else t for t in re.split(r"(\d+)", path.name)]


def pmc_traffic(workload_key: str, kernel_symbol: str = ""):
    """Bytes per force-kernel launch at the L2's fabric side, from the newest committed rocprofv3 --pmc passes
    (profiles/*pmc_force*.json: tools/profile_traffic.sh, older ones tools/pmc_summary.py) taken on THIS workload
    -- (dict, tag) or (None, None).  Reads are counted by request size: on gfx950 every fabric read request is
    128 B (TCC_EA0_RDREQ_128B = TCC_EA0_RDREQ: profiles/r04_v2_traffic.txt, calibrated on streams and gathers of
    known size), i.e. read bytes = 2 x FETCH_SIZE exactly -- the "1x" reading round 3 carried beside it is gone.
    The counters sit at the TCC's fabric side: Infinity-Cache hits are included (not HBM bytes proper; the
    requests cannot be told apart there)."""
    best = (None, None)
    for p in sorted((ROOT / "profiles").glob("*pmc_force*.json"), key=_by_age):      # rNN_vM names, oldest first
        try:
            d = json.loads(p.read_text())
        except Exception:
            continue
        sym = d.get("kernel_symbol", "").split("::")[-1]
        # (profiles of round 1-2 carry no symbol: they are the one-lane kernel's)
        if _profile_workload(p.name) == workload_key and (not kernel_symbol or not sym or sym == kernel_symbol):
            fetch, write = (d.get("fetch_size_kb_raw") or 0.0) * 1024.0, (d.get("write_size_kb") or 0.0) * 1024.0
            total = d.get("bytes_per_launch") or (2.0 * fetch + write)
            parts = None
            ab = d.get("ablation_read_bytes")
            if ab and d.get("read_bytes"):
                # where the reads come from: the diagnostics build's kernel with parts switched off
                full, goal, wall, pairs = (ab.get(k) for k in ("nothing switched off", "no goal stencil", "no wall term", "no pair work (phases 1-3)"))
                if full and goal and wall and pairs:
                    parts = {"potential_map_goal_stencil": full - goal, "distance_map_wall_term": full - wall,
                             "neighbour_gathers": full - pairs,
                             "own_state_and_index": full - (full - goal) - (full - wall) - (full - pairs),
                             "writes_state_keys_count_atomics": d.get("write_bytes")}
            best = ({"total": total, "read": d.get("read_bytes") or 2.0 * fetch, "write": d.get("write_bytes") or write, "parts": parts}, p.name)
    return best


# VALU issue on gfx950 (tools/microbench/valu_issue.hip, profiles/r02_valu_issue.txt): a SIMD-32
# issues one wave64 VALU instruction per 2 cycles at best (MI355X_MICROARCH.md); with the 6
# waves per SIMD the force kernel runs at, independent v_fma_f32 streams measured 2.3-2.4.
VALU_CYCLES_PER_INST = 2.0
N_SIMDS = 1024


def valu_floor(avg_launch_ms: float, workload_key: str, agents: float, kernel_symbol: str, agents_per_wave: int = 64):
    """Instruction-issue floor of the force kernel from the newest committed stall-counter profile
    of THIS workload taken on THE KERNEL THE RUN LAUNCHES (profiles/*_stalls.json, tools/profile_stalls.sh /
    profile_workload.sh; the symbol comes from pedoni_hip_force_kernel_info): VALU wave-instructions
    PER WAVE of the profiled launches x the waves of this run's launch x 2 cycles / 1024 SIMDs, at the
    clock the profiled launches held.  None when no profile of that kernel exists.  (Round 2 applied a
    1e6-agent profile's launch total to any run: 8x off at 8e6 agents.)"""
    best = None
    for p in sorted((ROOT / "profiles").glob("*_stalls.json"), key=_by_age):             # oldest first
        if _profile_workload(p.name) != workload_key or "_base_" in p.name:
            continue
        try:
            d = json.loads(p.read_text())
        except Exception:
            continue
        for k, v in d.items():
            if k.split("::")[-1] == kernel_symbol and v.get("SQ_INSTS_VALU") and v.get("SQ_WAVES"):
                best = (v, p.name, k)
    if not best:
        return None
    v, tag, symbol = best
    waves = float(int((agents + agents_per_wave - 1) // agents_per_wave))
    per_wave = v["SQ_INSTS_VALU"] / v["SQ_WAVES"]
    insts = per_wave * waves
    floor_cycles = insts * VALU_CYCLES_PER_INST / N_SIMDS
    clock_ghz = v.get("clock_ghz") or 2.1
    floor_ms = floor_cycles / (clock_ghz * 1e6)
    return {"insts_per_launch": insts, "insts_per_wave": per_wave, "waves": waves, "profile_waves": v["SQ_WAVES"],
            "cycles_per_inst": VALU_CYCLES_PER_INST, "simds": N_SIMDS, "floor_ms": floor_ms,
            "frac": floor_ms / avg_launch_ms, "clock_ghz": clock_ghz, "profile": tag, "kernel_symbol": symbol}


def self_launch(n: int) -> int:
    """`python bench.py --gpus N` outside torch.distributed.run: start the N ranks ourselves.
    This process has imported neither torch nor the HIP library and never touches the GPU; the
    ranks run under `python -m torch.distributed.run` as a CHILD process (subprocess, never an
    exec), on a free local port, with this command line's own arguments.  The child inherits fd 1,
    so rank 0's ONE JSON line goes straight through; its exit code is ours.  The watchdog lives in
    the ranks (a hung collective ends them in minutes, with the stage on stderr)."""
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
           "--master-addr", "127.0.0.1", "--master-port", str(port), str(Path(__file__).resolve()), *sys.argv[1:]]
    print(f"[bench] --gpus {n} without WORLD_SIZE: launching {n} ranks: {' '.join(cmd)}", file=sys.stderr, flush=True)
    env = dict(os.environ)
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 8) // n)))
    child = subprocess.Popen(cmd, env=env)
    try:
        return child.wait()
    except KeyboardInterrupt:
        child.terminate()
        return child.wait()


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
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # plain `python bench.py --gpus N`: this process becomes the launcher and nothing else
        sys.exit(self_launch(args.gpus))
    # RCCL prints a version banner on stdout: keep fd 1 for the ONE JSON line
    json_fd = os.dup(1)
    os.dup2(2, 1)
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        print(f"[bench] --gpus {args.gpus} but WORLD_SIZE={world}: running with {world} rank(s)", file=sys.stderr)
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
    ranks_seen = 1
    wd = None
    if world > 1 or force_sharded:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        wd = Watchdog(rank)
        wd.stage("torch.distributed init", float(os.environ.get("PEDONI_BENCH_INIT_BOUND_S", "240")))
        import torch.distributed as dist
        # "nccl" IS RCCL on ROCm; PEDONI_DIST_BACKEND=gloo only rehearses the code path
        backend = os.environ.get("PEDONI_DIST_BACKEND", "nccl")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)
        # small control tensors of the collectives: on the device for RCCL, on the host for a gloo rehearsal
        ctl = torch.device("cuda", local_rank) if backend == "nccl" else torch.device("cpu")
        seen = torch.ones(1, dtype=torch.int32, device=ctl)
        dist.all_reduce(seen)                       # the first collective: how many ranks really answer
        ranks_seen = int(seen.item())
    stream = torch.cuda.Stream()
    torch.cuda.set_stream(stream)   # kernels + collective on one non-default stream

    bound_scale = float(os.environ.get("PEDONI_BENCH_BOUND_SCALE", "1"))   # (tests shrink the bounds)

    def stage(name, bound_s):
        if wd is not None:
            wd.stage(name, bound_s * bound_scale)
            if os.environ.get("PEDONI_BENCH_HANG_AT") == name and rank == world - 1:
                time.sleep(10_000)                  # test hook: a rank that stops answering (tests/test_cli.py)

    G = args.gpus
    n_per = args.agents_per_gpu
    density = args.density
    side = float(np.sqrt(n_per / density))          # 1000 m at 1e6 agents and rho = 1
    width, height = side, side * G
    obstacles, waypoints = box_geometry(width, height)
    workload = (f"uniform crowd N={n_per * G:.0e} ({n_per:.0e}/GPU) in a {width:.0f}x{height:.0f} m box, "
                f"rho={density:g}/m^2, neighbor grid 1.4 m, field maps 0.25 m, fp32; "
                f"{SETTLE_TICKS} untimed ticks old at the warmup for every N").replace("e+0", "e")

    custom_crowd = None
    if args.workload != "c3":
        if G != 1:
            sys.exit("--workload c2/c4 are single-GPU configurations")
        obstacles, waypoints, (width, height), custom_crowd, workload = other_workload(args.workload)
        n_per = 100_000 if args.workload == "c2" else 1_000_000

    stage("field", 420.0)
    shm_dir = None
    if dist is not None and world > 1:
        field, t_field, field_how, shm_dir = shared_field(dist, torch, ctl, rank, (width, height), 0.25, obstacles, waypoints)
    else:
        t0 = time.perf_counter()
        field = host.Field.build((width, height), 0.25, obstacles, waypoints)
        t_field, field_how = time.perf_counter() - t0, "built in process"

    opt = abi.Options(math_mode=abi.MATH_FAST if args.math == "fast" else abi.MATH_EXACT,
                      gpu_work_size=args.work_size, initial_capacity=int(n_per * 1.3),
                      use_distance_map=args.workload != "c4seg")
    map_rows = bounds = None
    stage("model", 240.0)
    if G > 1 or force_sharded:
        from pedoni_amd.sharded import ShardedModel, band_rows, default_halo_cap
        rows = int(np.ceil(np.float32(height) / np.float32(opt.neighbor_grid_unit)))   # neighbor_grid.rs:14-20
        bounds = band_rows(rows, G)                # uniform crowd: equal rows = equal agents
        # each rank uploads only its band's texel rows of the three maps (1/G of 0.5 GB each at G = 8)
        map_rows = abi.shard_map_rows(bounds[rank], bounds[rank + 1], 0, opt.neighbor_grid_unit, field.unit,
                                      field.shape[0])

    def new_model():
        m = abi.HipModel(opt, (width, height), field.distance_map, field.potential_maps,
                         field.unit, obstacles, device=local_rank, map_rows=map_rows)
        if G > 1 or force_sharded:
            m.set_stream(stream.cuda_stream)
        return m

    model = new_model()
    sharded = G > 1 or force_sharded
    if sharded:
        lo, hi = bounds[rank], bounds[rank + 1]
        # this rank's agents: exactly its own band of grid rows (2 m clear of the outer walls)
        y_lo, y_hi = lo * 1.4 + 0.01, hi * 1.4 - 0.01
    else:
        y_lo, y_hi = 0.0, height
    if custom_crowd is not None:
        pos, dest, v0, vel = custom_crowd(field)
    else:
        pos, dest, v0, vel = uniform_crowd(
            n_per, (12.0, width - 12.0), (max(y_lo, 2.0), min(y_hi, height - 2.0)), seed=12345 + rank)

    exchange = None
    runner = shard = None
    verified = None
    crowd_age = 0          # ticks the crowd has been through before the warmup
    if sharded:
        assert model.neighbor_grid_shape()[0] == rows
        cap = default_halo_cap(int(width * 1.4 * density))

        def agreed(ok: int) -> bool:
            # (every rank takes part in every collective below whatever failed locally)
            flag = torch.tensor([ok], dtype=torch.int32, device=ctl)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            return int(flag.item()) == 1

        def torch_runner(m):
            r = ShardedModel(m, rank, G, dist, torch, halo_cap=cap, bounds=bounds,
                             overlap=os.environ.get("PEDONI_OVERLAP") == "1")
            assert (r.owner_of(pos[:, 1]) == rank).all()
            r.load(pos, dest, v0, vel)
            return r

        why = ""
        if os.environ.get("PEDONI_EXCHANGE", "rccl") == "rccl" and dist.get_backend() == "nccl":
            # the driver below the C-ABI: libpedoni_hip owns an RCCL communicator and sends /
            # receives the lists itself (ncclSend / ncclRecv with rank +- 1 on the model's
            # stream).  torch.distributed only carries the 128-byte id and the timing barrier.
            stage("rccl communicator + token ring", 240.0)
            ok = 1
            idt = torch.zeros(abi.SHARD_ID_BYTES, dtype=torch.uint8, device=ctl)
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
                # The direct path had never run with a neighbour before the driver's multi-GPU run
                # (one GPU per development box).  So it proves itself HERE, every run: 16 ticks of it
                # (4 plain + 12 overlapped in one call: the edge-first launch, the lists unpacked ahead on the
                # communication stream, and the one unpack in 8 that re-reads the live count on the model's
                # stream all take part) against 16 ticks of the torch all_gather driver on a second
                # model with the same crowd; every rank compares the two states bit for bit.
                stage("verify direct exchange against all_gather", 240.0)
                # (staged: after each stage every rank learns whether ALL ranks got through it -- a rank
                # that failed never leaves the others inside a collective it will not join; the
                # all-reduce of `agreed` is also the barrier that keeps the two communicators, the
                # library's and torch's, from ever being in flight on one device together)
                ref_model = None
                try:
                    model.append(pos, dest, v0, vel)
                    shard.begin()
                    shard.tick_n(4); shard.set_overlap(True); shard.tick_n(12); shard.set_overlap(False)
                    crowd_age += 16
                    torch.cuda.synchronize()
                except Exception as e:             # noqa: BLE001
                    ok, why = 0, str(e)
                if agreed(ok):
                    try:
                        ref_model = new_model()
                    except Exception as e:         # noqa: BLE001
                        ok, why = 0, str(e)
                    if agreed(ok):
                        try:
                            ref = torch_runner(ref_model)
                            ref.tick_n(16)
                            torch.cuda.synchronize()
                            a, b = model.download(), ref_model.download()
                            same = all(x.shape == y.shape and np.array_equal(x.view(np.uint32), y.view(np.uint32))
                                       for x, y in zip(a, b))
                            if not same:
                                ok, why = 0, "direct-exchange state differs from the all_gather driver's after 16 ticks"
                        except Exception as e:     # noqa: BLE001
                            ok, why = 0, str(e)
                    else:
                        ok = 0
                else:
                    ok = 0
                if ref_model is not None:
                    ref_model.close()
                ok = 1 if agreed(ok) else 0
                verified = bool(ok)
            if ok == 1:
                exchange = ("direct RCCL ncclSend/ncclRecv to rank+-1, driven by libpedoni_hip (pedoni_shard_tick_n); "
                            "verified in this run: bit-equal to the all_gather driver over 16 ticks (4 plain + 12 overlapped) on every rank")
            else:
                print(f"[bench] rank {rank}: direct RCCL path unavailable ({why or 'another rank failed'}); "
                      "falling back to torch.distributed all_gather", file=sys.stderr)
                if shard is not None:
                    shard.close()
                shard = None
                model.close()
                model = new_model()
                crowd_age = 0          # (a new model: the crowd starts over)
        if shard is None:
            stage("all_gather driver", 240.0)
            runner = torch_runner(model)
            exchange = "torch.distributed all_gather_into_tensor (RCCL), driven from Python"
            if why:
                exchange += f" (direct path refused: {why[:120]})"
    if shm_dir is not None:
        # every rank has uploaded its slices (two models at most): the files can go.  (Mapped pages
        # stay valid after the unlink; a watchdog exit can no longer leak 1.5 GB of /dev/shm.)
        dist.barrier()
        if rank == 0:
            import shutil
            shutil.rmtree(shm_dir, ignore_errors=True)

    if shard is not None:
        step_fn = shard.tick_n
        # plain tick (exchange, then the whole update) or overlapped (the next exchange under the
        # interior rows' update)?  Which is faster depends on what the exchange costs on this
        # node: time 16 ticks of each, every rank keeps the mode that was faster for the slowest.
        stage("tick-form probe", 240.0)
        mode_ms = {}
        for mode in (False, True):
            shard.set_overlap(mode)
            shard.tick_n(5)
            torch.cuda.synchronize(); dist.barrier(); torch.cuda.synchronize()   # (the library's exchanges drained before torch's collective starts)
            t0 = time.perf_counter()
            shard.tick_n(16)
            crowd_age += 21
            torch.cuda.synchronize()
            t = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=ctl)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            mode_ms[mode] = float(t.item()) / 16 * 1e3
        use_overlap = mode_ms[True] < mode_ms[False]
        shard.set_overlap(use_overlap)
        exchange += (f"; tick form: {'overlapped' if use_overlap else 'plain'} "
                     f"(probe: plain {mode_ms[False] * 1e3:.0f} us, overlapped {mode_ms[True] * 1e3:.0f} us per tick; "
                     f"verification + probe = {crowd_age} ticks before the warmup)")
    elif runner is not None:
        step_fn = runner.tick_n
    else:
        model.append(pos, dest, v0, vel)
        step_fn = model.tick_n

    # Every configuration times a crowd of the same age.  The N > 1 runs have ticked theirs already
    # (verification + tick-form probe); all runs tick on, untimed, to SETTLE_TICKS before the warmup.
    # It matters: the crowd starts as independent uniform positions (agents centimetres apart), the
    # first ~100 ticks push those apart and the tick gets ~10 % cheaper on the way
    # (tools/short_run_cost.py) -- a 1-GPU line on a fresh crowd beside N-GPU lines on a 58-tick-old
    # one would flatter the scaling curve.
    if crowd_age < SETTLE_TICKS:
        stage("settling the crowd", 60.0 + 0.05 * (SETTLE_TICKS - crowd_age) * max(1.0, n_per / 1e6))
        remaining = SETTLE_TICKS - crowd_age
        # (the single-GPU tick replays captured runs of 16 / 8 / 4 / 2 ticks: settle in calls of every run length, so that
        # every graph the timed region may want is captured and instantiated here, not there)
        for chunk in ([16, 16, 8, 8, 4, 4, 2] if remaining == 58 and not sharded else [remaining]):
            step_fn(chunk)
        crowd_age = SETTLE_TICKS

    def barrier():
        # (drain first: in the overlapped form an exchange of the library's own communicator may still be
        # in flight, and two communicators are never given work on one device at the same time)
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    # bounds of the timed stages: generous multiples of what the ticks should take (a tick of 1e6
    # agents is ~0.12 ms; 50 ms per tick + a minute is two orders of magnitude of slack)
    tick_bound = lambda k: 60.0 + 0.05 * k * max(1.0, n_per / 1e6)
    stage("warmup", tick_bound(args.warmup))
    step_fn(args.warmup)
    barrier()
    n_before = model.owned_count() if sharded else model.get_pedestrian_count()
    # inside the timed region the dominant kernel is event-timed on every `every`-th tick only
    # (timed ticks launch eagerly inside one hipEvent pair, the others replay captured runs of
    # ticks): every 17th / 9th tick on long runs (the 16 / 8 ticks between two timed ones replay as
    # one graph launch); the driver's short runs time the first 7 ticks of the region (below);
    # the per-kernel pass after the timed region times 20 more, all kernels
    # (the library replays runs of up to 16 plain ticks from ONE captured graph -- the stream idles ~8 us between two
    # graph launches -- so the period is one more than a run length it has: 17 or 9)
    every = 17 if args.steps >= 170 else (9 if args.steps >= 90 else 3)
    # short runs (the driver's 20 steps): the timed launches are the region's FIRST 7 ticks in a row, the rest of
    # the region replays in long runs -- a timed tick between every two plain ones meant a graph launch (and its
    # ~8 us of idle stream) per pair of ticks
    burst = 7 if args.steps < 90 and args.steps >= 14 else 0
    if burst:
        every = max(args.steps, burst)
    if not args.no_profile:
        model.profile(True, kernels=[abi.K_FORCE], every=every, burst=burst)
        model.kernel_times(reset=True)
    stage("timed region", tick_bound(args.steps))
    barrier()
    t0 = time.perf_counter()
    step_fn(args.steps)
    barrier()
    elapsed = time.perf_counter() - t0
    stage("per-kernel pass + reduction", tick_bound(20) + 60.0)
    ktimes = model.kernel_times(reset=True) if not args.no_profile else {}
    model.profile(False)
    n_after = model.owned_count() if sharded else model.get_pedestrian_count()
    breakdown, force_pass = {}, None
    if not args.no_profile:
        model.profile(True)
        step_fn(min(args.steps, 20))
        barrier()
        bt = model.kernel_times(reset=True)
        model.profile(False)
        breakdown = {k: v["total_ms"] / max(v["launches"], 1) * (v["launches"] / min(args.steps, 20))
                     for k, v in bt.items() if v["launches"]}
        force_pass = bt.get("force_integrate")

    agents_local = 0.5 * (n_before + n_after)       # despawns during the run are negligible
    if dist is not None and world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=ctl)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        a = torch.tensor([agents_local], dtype=torch.float64, device=ctl)
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
                       "math_mode": args.math, "crowd_age_at_warmup_ticks": crowd_age,
                       "parallelism": (f"row-bands x{G} ({ranks_seen} ranks answered the first all-reduce); {exchange}"
                                       if sharded else "single GPU"),
                       "field_build_s": round(t_field, 2), "field": field_how,
                       "tick_algorithmic_GBps": BYTES_TICK * value / 1e9},
        }
        if verified is not None:
            out["config"]["direct_exchange_verified"] = verified
        fk = ktimes.get("force_integrate")
        if fk and fk["launches"]:
            avg_ms = fk["total_ms"] / fk["launches"]
            achieved = BYTES_FORCE * agents_local / (avg_ms * 1e-3) / 1e9
            ksym, per_wave = model.force_kernel_info(int(agents_local))
            traffic, traffic_tag = pmc_traffic(args.workload, ksym) if G == 1 and n_per in (100_000, 1_000_000) else (None, None)
            valu = valu_floor(avg_ms, args.workload, agents_local, ksym, per_wave) if G == 1 else None
            hbm_frac = achieved / HBM_PEAK_GBS
            out["roofline"] = {
                # the roof the kernel is closer to: HBM bytes at 8 TB/s, or VALU issue at one
                # wave instruction per 2 cycles per SIMD (no MFMA on this path)
                "bound": "valu" if valu and valu["frac"] > hbm_frac else "hbm",
                "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": hbm_frac,
                # bytes per launch at the TCC's fabric side in the committed --pmc profile of this workload and
                # kernel (see pmc_traffic): one number -- read requests are counted by size -- with its parts
                "traffic": traffic["total"] if traffic else None,
                "traffic_read": traffic["read"] if traffic else None,
                "traffic_write": traffic["write"] if traffic else None,
                "traffic_parts": traffic["parts"] if traffic else None,
                "traffic_profile": traffic_tag,
                "traffic_note": ("from the committed rocprofv3 --pmc profile of this workload, not measured in this run; fabric-side "
                                 "bytes, Infinity-Cache hits included (not HBM bytes proper); reads = 128 B x TCC_EA0_RDREQ (every read "
                                 "request is 128 B on gfx950: profiles/r04_v2_traffic.txt); compulsory part beyond the algorithmic 40 B / "
                                 "agent: the 64 MB potential map, which a crowd at 1 agent / m^2 touches in full each tick") if traffic else None,
                "kernel": "force_integrate", "kernel_symbol": ksym, "avg_launch_ms": avg_ms,
                "timed_launches": fk["launches"],
                "algorithmic_bytes_per_launch": BYTES_FORCE * agents_local,
                "valu": valu,
            }
            if force_pass and force_pass["launches"]:
                # the per-kernel pass right after the timed region: 20 more launches of the same kernel
                out["roofline"]["avg_launch_ms_kernel_pass"] = force_pass["total_ms"] / force_pass["launches"]
                out["roofline"]["kernel_pass_launches"] = force_pass["launches"]
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
            fm.tick_n(crowd_age + args.warmup)
            fm.synchronize()
            nb = fm.get_pedestrian_count()
            fm.profile(True, kernels=[abi.K_FORCE], every=every, burst=burst)
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
            # BASELINE.md holds no published number for this metric (the reference publishes none):
            # vs_baseline stays null; the ratio to the CPU port timed in this very run is its own field
            out["vs_cpu_baseline"] = value / out["cpu_baseline"]["value"]
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(out) + "\n").encode())

    stage("teardown", 120.0)
    if shard is not None:
        shard.close()                              # ncclCommDestroy before the model goes
    model.close()
    if dist is not None:
        dist.destroy_process_group()
    if wd is not None:
        wd.done()


if __name__ == "__main__":
    main()
