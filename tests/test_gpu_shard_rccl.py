"""pedoni_shard_tick_n -- the driver `bench.py --gpus N` runs -- with world > 1 on one GPU.

A 1-GPU box cannot bring up a real RCCL group of more than one rank (RCCL refuses two ranks on
one device), so until the driver's 8-GPU run the rank+-1 exchange, the overlapped form on its own
stream and the re-cut's all-reduce + bulk exchange had never executed with a neighbour.  Here
every rank is a host thread of ONE child process with its own model and stream, and librccl is
replaced by tests/loopback_rccl (event-ordered device copies; PEDONI_RCCL_LIB): every line of OUR
side of the protocol runs -- buffers, offsets, peers, streams, group bracketing, message sizes --
and the merged bands must equal the unsharded model bit for bit.  What this cannot cover is RCCL
itself and the wire."""
import json
import os
import subprocess
import sys
from pathlib import Path

import pytest

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent
LOOPBACK = ROOT / "tests" / "loopback_rccl" / "libloopback_rccl.so"


def _run(world, mode, extra_env=None):
    assert LOOPBACK.exists(), "tests/loopback_rccl is not built (python -m pedoni_amd.build)"
    env = dict(os.environ, PEDONI_RCCL_LIB=str(LOOPBACK), LOOPBACK_RCCL_TIMEOUT_S="30")
    env.update(extra_env or {})
    p = subprocess.run([sys.executable, str(ROOT / "tests" / "loopback_shard_runner.py"), str(world), mode],
                       env=env, capture_output=True, text=True, timeout=600)
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert p.returncode == 0 and lines, f"runner failed ({p.returncode}):\n{p.stdout[-2000:]}\n{p.stderr[-3000:]}"
    return json.loads(lines[-1])


@pytest.mark.parametrize("world,mode", [(2, "plain"), (4, "overlap"), (3, "recut"), (5, "recut_overlap")])
def test_rccl_driven_ticks_equal_the_single_model_bitwise(hip, world, mode):
    out = _run(world, mode)
    assert out["ok"], out
    assert out["count_equal"] and out["dest_equal"] and out["bit_equal"], out
    # every tick every interior boundary carries one send and one receive each way (+ the token ring)
    per_tick = 2 * (world - 1)
    assert out["sends"] == out["recvs"] >= per_tick * (out["ticks"] + 1), out
    if mode.startswith("recut"):
        assert out["allreduces"] >= world * (out["ticks"] // 4), out       # one histogram all-reduce per re-cut and rank
        assert out["bounds1"] != out["bounds0"], "the bands were never re-cut"
        assert out["loads"] != out["loads0"]            # rows (and their agents) changed owner over the wire
    else:
        assert out["allreduces"] == 0 and out["bounds1"] == out["bounds0"], out


@pytest.mark.parametrize("world,mode,env,form", [
    (3, "overlap", {"PEDONI_FORCE_GROUP": "1"}, "edge_first"),                                       # default build, 8 slots
    (4, "recut_overlap", {"PEDONI_FORCE_GROUP": "1", "PEDONI_FORCE_KERNEL": "s94:6"}, "edge_first"),  # the 7-wave build
    (2, "overlap", {"PEDONI_FORCE_GROUP": "1", "PEDONI_FORCE_KERNEL": "s94:6", "LOOPBACK_MATH": "fast"}, "edge_first"),
    (3, "overlap", {"PEDONI_FORCE_GROUP": "1", "PEDONI_SHARD_FORM": "split"}, "split"),
    (4, "overlap", {}, "split"),                                                                      # small bands: 2-4 lanes per agent
    (2, "overlap", {"LOOPBACK_BIG": "1"}, "edge_first"),              # ~5e5 agents per band: the kernel the by-size rule picks, 30 ticks
])
def test_every_form_of_the_overlapped_tick_equals_the_single_model(hip, world, mode, env, form):
    """The overlapped tick has two forms.  Edge-first: ONE force launch whose first workgroups take the
    edge rows and release the exchange from inside the launch (a word in device memory, polled by a wave
    on the communication stream) -- the form large bands run, forced here onto these small ones by asking
    for the one-lane-per-agent kernels.  Split: an edge launch, then the interior launch.  Both, with 2-4
    bands and real neighbours, bit-equal to the unsharded model; and the form asked for is the one that ran."""
    out = _run(world, mode, env)
    assert out["ok"], out
    assert out["count_equal"] and out["dest_equal"] and out["bit_equal"], out
    other = "split" if form == "edge_first" else "edge_first"
    for f in out["forms"]:
        assert f[form] >= 7 and f[other] == 0, out["forms"]        # (a re-cut tick is a plain one)


@pytest.mark.parametrize("world,env,min_rows", [
    # a straggling neighbour: every list reaches the receiver's communication stream 400 us late, i.e. WHILE the
    # next tick's scan is already spinning on the "lists unpacked" word (the riding wait's slow path)
    (3, {"PEDONI_FORCE_GROUP": "1", "LOOPBACK_RCCL_RECV_DELAY_US": "400"}, 0),
    # the same with the wait taken out of the scan's workgroups into one wave ahead of it (what a band of more
    # rows than half the chip's resident workgroups gets: run_row_scan)
    (3, {"PEDONI_FORCE_GROUP": "1", "LOOPBACK_RCCL_RECV_DELAY_US": "400", "PEDONI_SCAN_WAIT_ROWS_MAX": "0"}, 0),
    # bands of MORE grid rows than the chip holds scan workgroups (2048), late lists: one
    # spinning workgroup per row would take every wave slot and keep out the unpack that stores the word
    (2, {"PEDONI_FORCE_GROUP": "1", "LOOPBACK_RCCL_RECV_DELAY_US": "400", "LOOPBACK_TALL": "1"}, 4400),
])
def test_overlapped_tick_with_late_lists_and_bands_taller_than_the_chip(hip, world, env, min_rows):
    out = _run(world, "overlap", env)
    assert out["ok"], out
    assert out["count_equal"] and out["dest_equal"] and out["bit_equal"], out
    assert out["grid_rows"] >= min_rows
    if min_rows:
        rows = [b - a for a, b in zip(out["bounds0"], out["bounds0"][1:])]
        assert max(rows) > 2048, rows
    for f in out["forms"]:
        assert f["edge_first"] >= 7, out["forms"]


def test_a_failed_send_inside_a_group_leaves_no_group_open(hip):
    """NCCL_TRY used to return from inside an open ncclGroupStart.  With the
    first ncclSend made to fail, the entry point reports it, the thread's group depth is back to
    zero, and the very next exchange works."""
    out = _run(1, "fault", {"LOOPBACK_RCCL_FAIL_SEND": "0"})
    assert out["first_error"] and "ncclSend" in out["first_error"], out
    assert out["depth_after_failure"] == 0, out
    assert out["second"] is True, out


@pytest.mark.parametrize("overlap", ["0", "1"])
def test_a_tick_that_fails_midway_stops_the_shard(hip, overlap):
    """an error inside a band's tick (here: an ncclSend of the third tick fails on one rank; its
    neighbour runs into the receive's bound) used to leave `unpacked_ahead` / `in_flight` as they were and the
    next tick skipped its unpack or exchanged twice.  Now the failing call reports the error, settles what is
    under way and the shard refuses further ticks until pedoni_shard_begin."""
    # sends so far: the token ring (2 per interior boundary and direction) and begin's pack; the 9th send is inside a tick
    out = _run(2, "tick_fault", {"LOOPBACK_RCCL_FAIL_SEND": "9", "LOOPBACK_RCCL_TIMEOUT_S": "4", "LOOPBACK_OVERLAP": overlap})
    assert out["ok"], out
    assert any(f != "ok" for f in out["first"]), out                 # the failure was reported ...
    failed = [r for r, f in enumerate(out["first"]) if f != "ok"]
    for r in failed:
        assert "pedoni_shard_begin" in out["first"][r], out            # ... says what to do ...
        assert out["second"][r] != "ok" and "pedoni_shard_begin" in out["second"][r], out   # ... and the shard stays stopped
