"""Batch-of-frames mode over several ranks, on CPU: the frame sharding arithmetic, the
launcher bench.py uses when no launcher sits above it, and the host-side group (rendezvous file + TCP
star) that carries the RCCL id, the barrier and the max-over-ranks timing -- world_size-2 processes,
each result cross-checked against the same collective done with torch.distributed's gloo backend."""
import json
import os
import socket
import subprocess
import sys
import textwrap

import numpy as np
import pytest

from transflow_amd.batch import batch_starts, frames_needed, gather_calls, gather_counts, pass_pairs, shard_range

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("total,world", [(256, 8), (257, 8), (255, 8), (7, 8), (0, 3), (10, 1), (64, 2)])
def test_shard_range_partitions_exactly(total, world):
    spans = [shard_range(total, r, world) for r in range(world)]
    assert spans[0][0] == 0 and spans[-1][1] == total
    for (a0, a1), (b0, b1) in zip(spans, spans[1:]):
        assert a1 == b0 and a0 <= a1
    sizes = [b - a for a, b in spans]
    assert max(sizes) - min(sizes) <= 1 and sum(sizes) == total
    for a, b in spans:                          # pair t needs frames t and t+1: one-frame halo
        assert frames_needed((a, b)) == ((a, b + 1) if b > a else (a, a))
    with pytest.raises(ValueError):
        shard_range(4, 4, 4)


@pytest.mark.parametrize("n,batch", [(255, 16), (32, 16), (31, 16), (16, 16), (5, 16), (0, 16), (33, 8), (17, 16)])
def test_batch_starts_cover_the_shard(n, batch):
    starts = batch_starts(n, batch)
    covered = set()
    for s in starts:
        size = min(batch, n)
        assert 0 <= s and s + size <= n
        covered.update(range(s, s + size))
    assert covered == set(range(n))
    assert len(starts) == (0 if n == 0 else max(1, -(-n // batch)))


def replay_gather(calls):
    """csrc/batch.hip's posting rules for ONE tf_batch_gather, replayed on the argument lists the ranks pass:
    a non-root rank posts one Send of send_bytes (none when it is 0); the root posts, for every rank r != root with
    recv_bytes[r] > 0, one Recv of recv_bytes[r] at recv_dev + sum(recv_bytes[:r]), and copies its own send_bytes
    locally (which must equal recv_bytes[root]).  Returns the byte ranges written on the root; raises if a Recv
    has no Send, a Send has no Recv, their sizes differ, or a range leaves the receive buffer."""
    root = calls[0]["root"]
    rc = calls[root]
    assert rc["rank"] == root and rc["recv_bytes"] is not None
    counts = rc["recv_bytes"]
    assert len(counts) == len(calls)
    assert counts[root] == rc["send_bytes"], "root's own count must equal what it sends (TF_ERR_ARG otherwise)"
    sends = {c["rank"]: c["send_bytes"] for c in calls if c["rank"] != root and c["send_bytes"] > 0}
    ranges, off = [], 0
    for r, n in enumerate(counts):
        lo, off = off, off + n
        if n == 0:
            continue
        assert off <= rc["recv_capacity"], f"rank {r}'s bytes leave the receive buffer"
        if r != root:
            assert r in sends, f"the root waits for rank {r}, which sends nothing: deadlock"
            assert sends.pop(r) == n, f"rank {r} sends a different size than the root expects"
        ranges.append((r, lo, off))
    assert not sends, f"ranks {sorted(sends)} send but the root posts no receive: deadlock"
    return ranges


@pytest.mark.parametrize("world", [1, 2, 3, 4, 7, 8])
@pytest.mark.parametrize("clip_frames", [256, 257, 40])
@pytest.mark.parametrize("batch,equal", [(32, False), (7, False), (32, True)])
def test_gather_call_sequence_has_a_send_for_every_receive(world, clip_frames, batch, equal):
    """The sequence of (rank, send_bytes, recv_bytes[]) bench.py's gather leg passes to tf_batch_gather: with
    T = 256 over 8 ranks seven ranks hold 32 pairs per pass and the last one 31, and a gather per image (round 2)
    left the root waiting for a 32nd send rank 7 never made."""
    sys.path.insert(0, ROOT)
    import bench
    frame_bytes = 3840 * 2160 * 3
    plans = [bench.make_plan(clip_frames, batch, r, world, equal) for r in range(world)]
    per_pass = [p["pairs_per_pass"] for p in plans]
    assert per_pass == pass_pairs(clip_frames - 1, batch, world, equal)
    if equal:
        assert len(set(per_pass)) == 1 and per_pass[0] == min(min(batch, p["n_pairs"]) for p in plans)
    else:
        assert per_pass == [min(batch, p["n_pairs"]) for p in plans]
    for p in plans:                                        # every pass lies inside the rank's shard
        for s0 in p["pass_starts"]:
            assert 0 <= s0 and s0 + p["pairs_per_pass"] <= p["n_pairs"]
    calls = gather_calls(per_pass, frame_bytes)
    assert [c["rank"] for c in calls] == list(range(world))  # one call per rank and pass: the same number everywhere
    ranges = replay_gather(calls)
    counts, offsets = gather_counts(per_pass, frame_bytes)
    assert [(r, lo, hi) for r, lo, hi in ranges] == [(r, offsets[r], offsets[r] + counts[r]) for r in range(world) if counts[r]]
    assert sum(hi - lo for _, lo, hi in ranges) == sum(per_pass) * frame_bytes == calls[0]["recv_capacity"]
    for (_, _, hi), (_, lo, _) in zip(ranges, ranges[1:]):
        assert hi == lo                                     # frames of all ranks side by side, no gap, no overlap


def replay_gather_at(calls):
    """csrc/batch.hip's rules for ONE tf_batch_gather_at: as replay_gather, but rank r's bytes land at
    recv_offsets[r]; the root refuses ranges that leave the buffer or overlap (TF_ERR_ARG)."""
    root = calls[0]["root"]
    rc = calls[root]
    counts, offs = rc["recv_bytes"], rc["recv_offsets"]
    assert len(counts) == len(offs) == len(calls) and counts[root] == rc["send_bytes"]
    sends = {c["rank"]: c["send_bytes"] for c in calls if c["rank"] != root and c["send_bytes"] > 0}
    ranges = []
    for r, (n, at) in enumerate(zip(counts, offs)):
        if n == 0:
            continue
        assert at + n <= rc["recv_capacity"], f"rank {r}'s bytes leave the receive buffer"
        if r != root:
            assert r in sends, f"the root waits for rank {r}, which sends nothing: deadlock"
            assert sends.pop(r) == n
        ranges.append((r, at, at + n))
    assert not sends, f"ranks {sorted(sends)} send but the root posts no receive: deadlock"
    srt = sorted(ranges, key=lambda t: t[1])
    for (_, _, hi), (_, lo, _) in zip(srt, srt[1:]):
        assert hi <= lo, "ranges overlap"
    return ranges


@pytest.mark.parametrize("world", [1, 2, 3, 4, 7, 8])
@pytest.mark.parametrize("clip_frames", [256, 257, 40, 9])
@pytest.mark.parametrize("batch", [128, 32, 7])
def test_flows_to_root_calls_place_every_pair_of_the_clip_once(world, clip_frames, batch):
    """SURVEY 8e mode F: one tf_batch_gather_at per pass index; after the last one the root's buffer holds the flow of
    every pair of the clip at its clip position (a rank's last pass may repeat pairs of the one before: same place,
    same flow), every receive has its send, and a rank that has run out of passes sends nothing."""
    sys.path.insert(0, ROOT)
    import bench
    from transflow_amd.batch import flows_to_root_calls
    fb = 64                                                    # bytes per flow, small: positions are what is checked
    plans = [bench.make_plan(clip_frames, batch, r, world) for r in range(world)]
    if any(p["pairs_per_pass"] < 1 for p in plans):
        pytest.skip("the clip does not reach every rank (bench.py refuses such a launch)")
    calls = flows_to_root_calls(plans, fb)
    assert len(calls) == max(len(p["pass_starts"]) for p in plans)
    placed = {}
    for k, call in enumerate(calls):
        assert [c["rank"] for c in call] == list(range(world))
        for r, lo, hi in replay_gather_at(call):
            p = plans[r]
            assert k < len(p["pass_starts"]) and hi - lo == p["pairs_per_pass"] * fb
            first = p["pairs"][0] + p["pass_starts"][k]
            assert lo == first * fb
            for j in range(p["pairs_per_pass"]):
                assert placed.setdefault(first + j, r) == r    # a pair only ever comes from the rank that owns it
        for c in call:
            p = plans[c["rank"]]
            assert (c["send_bytes"] > 0) == (k < len(p["pass_starts"]))
    assert sorted(placed) == list(range(clip_frames - 1))
    assert calls[0][0]["recv_capacity"] == (clip_frames - 1) * fb


def test_replay_catches_the_round2_deadlock():
    """The per-image form of round 2 (equal counts assumed): on the 32nd image rank 7 has nothing to send."""
    per_pass = pass_pairs(255, 32, 8)
    assert per_pass == [32] * 7 + [31]
    nb = 100
    with pytest.raises(AssertionError, match="deadlock"):
        for i in range(max(per_pass)):
            calls = [{"rank": r, "root": 0, "send_bytes": nb if i < per_pass[r] else 0,
                      "recv_bytes": [nb] * 8 if r == 0 else None, "recv_capacity": 8 * nb} for r in range(8)]
            replay_gather(calls)


WORKER = textwrap.dedent("""
    import os, sys, json
    import numpy as np
    sys.path.insert(0, %r)
    from transflow_amd.batch import HostGroup, shard_range
    import torch, torch.distributed as dist
    g = HostGroup()
    assert g.world == 2
    dist.init_process_group("gloo", rank=g.rank, world_size=g.world)      # the cross-check transport
    uid = g.broadcast(bytes(range(128)) if g.rank == 0 else None)         # what carries the RCCL id
    t = torch.tensor(list(range(128)) if g.rank == 0 else [0] * 128, dtype=torch.uint8)
    dist.broadcast(t, src=0)
    assert bytes(t.tolist()) == uid
    lo, hi = shard_range(255, g.rank, g.world)                            # this rank's frame pairs
    g.barrier()
    elapsed = g.max_over_ranks(1.0 + g.rank)                              # timing = slowest rank
    tm = torch.tensor([1.0 + g.rank], dtype=torch.float64)
    dist.all_reduce(tm, op=dist.ReduceOp.MAX)
    assert elapsed == float(tm.item())
    total = g.sum_over_ranks(hi - lo)
    ts = torch.tensor([float(hi - lo)], dtype=torch.float64)
    dist.all_reduce(ts, op=dist.ReduceOp.SUM)
    assert total == float(ts.item())
    rows = g.gather({"rank": g.rank, "span": [lo, hi]})
    everyone = g.allgather(g.rank * 10)
    out = {"rank": g.rank, "uid_ok": uid == bytes(range(128)), "span": [lo, hi], "elapsed": elapsed, "total": total,
           "gathered": rows, "everyone": everyone}
    print("RESULT " + json.dumps(out), flush=True)
    g.close()
    dist.destroy_process_group()
""")


def test_two_rank_host_group_agrees_with_gloo(tmp_path):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    script = tmp_path / "worker.py"
    script.write_text(WORKER % ROOT)
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), TF_BATCH_RDZV=str(tmp_path / "rdzv"))
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.PIPE, text=True))
    results = {}
    for p in procs:
        out, err = p.communicate(timeout=180)
        assert p.returncode == 0, err[-2000:]
        line = [l for l in out.splitlines() if l.startswith("RESULT ")][0]
        r = json.loads(line[7:])
        results[r["rank"]] = r
    assert results[0]["uid_ok"] and results[1]["uid_ok"]
    assert results[0]["span"] == [0, 128] and results[1]["span"] == [128, 255]
    assert results[0]["elapsed"] == results[1]["elapsed"] == 2.0
    assert results[0]["total"] == results[1]["total"] == 255
    assert results[0]["gathered"] == [{"rank": 0, "span": [0, 128]}, {"rank": 1, "span": [128, 255]}]
    assert results[1]["gathered"] is None
    assert results[0]["everyone"] == results[1]["everyone"] == [0, 10]
    assert not os.path.exists(tmp_path / "rdzv")          # rank 0 removes the rendezvous file once everyone is in


@pytest.mark.parametrize("world", [2, 8])
def test_bench_launches_its_own_ranks_and_shards_the_clip(world):
    """`python bench.py --gpus N` with no launcher above it: N fresh rank processes, the T=256 clip's 255
    pairs sharded with a one-frame halo (BASELINE configs[4]); --dry-run stops before any GPU call."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "TF_BATCH_RDZV")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(world), "--dry-run"], env=env,
                         capture_output=True, text=True, timeout=180)
    assert out.returncode == 0, out.stderr[-2000:]
    d = json.loads(out.stdout.strip().splitlines()[-1])
    assert d["n_gpus"] == world and len(d["plans"]) == world
    # the dry-run line carries the keys of a measured line (the driver's contract + this bench's own), null where a GPU
    # would have spoken, and the workload's configuration as the measured line states it
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline", "parity_gate", "rccl_ranks", "gather",
                "flows_to_root", "per_rank_frames_per_s", "kernels_ms_per_step"):
        assert key in d, key
    # passes capped at the ranks' shards: a step covers the whole clip at every N from 2 up (the total is fixed: strong)
    assert d["value"] is None and d["scaling"] == "strong" and d["config"]["clip_frames"] == 256
    assert d["config"]["frame_pairs_per_step"] == sum(p["pairs_per_pass"] for p in d["plans"])
    pairs = [tuple(p["pairs"]) for p in d["plans"]]
    assert pairs == [shard_range(255, r, world) for r in range(world)]
    for r, p in enumerate(d["plans"]):
        assert p["rank"] == r
        assert tuple(p["frames"]) == (p["pairs"][0], p["pairs"][1] + 1)        # the halo frame
        assert p["pairs_per_pass"] == min(128, p["n_pairs"])                       # bench.py's default --batch
        assert p["pass_starts"] == batch_starts(p["n_pairs"], 128)
    if world == 8:                                                               # the case that must not hang
        assert [p["pairs_per_pass"] for p in d["plans"]] == [32] * 7 + [31]
        out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--dry-run", "--equal-batches"],
                             env=env, capture_output=True, text=True, timeout=180)
        assert out.returncode == 0, out.stderr[-2000:]
        d = json.loads(out.stdout.strip().splitlines()[-1])
        assert [p["pairs_per_pass"] for p in d["plans"]] == [31] * 8


def test_bench_under_torch_distributed_run_as_the_driver_launches_it():
    """`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py
    --gpus N ...`: the ranks exist already, find each other through the agent's pid + MASTER_PORT (no torch import
    in bench.py) and shard the clip; --dry-run stops before any GPU call."""
    pytest.importorskip("torch")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "TF_BATCH_RDZV")}
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                          "--master-addr", "127.0.0.1", "--master-port", "29541", os.path.join(ROOT, "bench.py"),
                          "--gpus", "2", "--steps", "3", "--warmup", "1", "--dry-run"], env=env, capture_output=True,
                         text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    d = json.loads([ln for ln in out.stdout.strip().splitlines() if ln.startswith("{")][-1])
    assert d["n_gpus"] == 2 and [tuple(p["pairs"]) for p in d["plans"]] == [shard_range(255, r, 2) for r in range(2)]


def test_bench_refuses_a_world_that_is_not_its_gpus():
    env = dict(os.environ, RANK="0", LOCAL_RANK="0", WORLD_SIZE="2")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--dry-run"], env=env,
                         capture_output=True, text=True, timeout=60)
    assert out.returncode != 0 and "WORLD_SIZE" in out.stderr


def test_launcher_stops_the_other_ranks_when_one_fails(tmp_path):
    from transflow_amd.batch import launch_ranks
    script = tmp_path / "r.py"
    script.write_text("import os, sys, time\nif os.environ['RANK'] == '1':\n    sys.exit(7)\ntime.sleep(60)\n")
    import time
    t0 = time.monotonic()
    rc = launch_ranks([sys.executable, str(script)], 3)
    assert rc == 15 or rc == 7        # 7 from the failing rank (or 15: SIGTERM's code of a terminated peer, abs())
    assert time.monotonic() - t0 < 30


def test_host_group_ignores_a_stranger_and_parses_nothing_it_sends(tmp_path):
    """The rendezvous file is private (0600, created exclusively) and carries two tokens: a connection that does not
    present the first is dropped before a byte of it is parsed; messages are JSON, never pickles."""
    import stat
    import threading
    import time
    from transflow_amd.batch import HostGroup
    path = str(tmp_path / "rdzv")
    box = {}

    def rank0():
        box["g"] = HostGroup(0, 2, path=path, timeout=60)

    t = threading.Thread(target=rank0)
    t.start()
    deadline = time.monotonic() + 30
    while not os.path.exists(path) and time.monotonic() < deadline:
        time.sleep(0.01)
    st = os.stat(path)
    assert stat.S_IMODE(st.st_mode) == 0o600
    where, hello, answer = open(path).read().split()
    host, port = where.rsplit(":", 1)
    import pickle
    with socket.create_connection((host, int(port))) as bad:       # a stranger: wrong token, then a pickle
        bad.sendall(b"x" * len(hello) + b"\x01\x00\x00\x00" + pickle.dumps({"boom": 1}))
        bad.settimeout(10)
        try:
            assert bad.recv(64) == b""                               # dropped without an answer
        except ConnectionResetError:
            pass                                                     # ... or reset: its bytes were never read
    g1 = HostGroup(1, 2, path=path, timeout=60)
    t.join(60)
    g0 = box["g"]
    res = {}
    th = threading.Thread(target=lambda: res.setdefault("r0", g0.allgather({"id": bytes(range(4)), "x": 1.5})))
    th.start()
    r1 = g1.allgather({"id": b"\xff", "x": None})
    th.join(30)
    assert r1 == res["r0"] == [{"id": bytes(range(4)), "x": 1.5}, {"id": b"\xff", "x": None}]
    with pytest.raises(TypeError):
        g1.gather(object())                                          # nothing but plain data travels
    g0.close()
    g1.close()


def test_host_group_takes_a_rank_that_comes_again_and_carries_plain_data_only(tmp_path):
    """A rank whose first connection failed on ITS side after rank 0 had accepted it (a timeout, a reset) retries with the
    right token: its new connection takes the old one's place instead of being turned away until both give up.  Messages:
    dict keys must be strings (JSON would silently turn others into strings), an array is not a scalar."""
    import struct
    import threading
    import time
    import numpy as np
    from transflow_amd.batch import HostGroup, _encode
    path = str(tmp_path / "rdzv")
    box = {}
    t = threading.Thread(target=lambda: box.setdefault("g", HostGroup(0, 3, path=path, timeout=60)))
    t.start()
    deadline = time.monotonic() + 30
    while not os.path.exists(path) and time.monotonic() < deadline:
        time.sleep(0.01)
    where, hello, answer = open(path).read().split()
    host, port = where.rsplit(":", 1)
    first = socket.create_connection((host, int(port)))                # rank 1's first attempt: accepted ...
    first.sendall(hello.encode() + struct.pack("<i", 1))
    assert first.recv(len(answer)).decode() == answer
    first.close()                                                      # ... then lost on rank 1's side
    g1 = HostGroup(1, 3, path=path, timeout=60)                        # it comes again
    g2 = HostGroup(2, 3, path=path, timeout=60)
    t.join(60)
    g0 = box["g"]
    res = {}
    th = [threading.Thread(target=lambda g=g, k=k: res.setdefault(k, g.allgather(k))) for g, k in ((g0, "a"), (g2, "c"))]
    for x in th:
        x.start()
    assert g1.allgather("b") == ["a", "b", "c"]
    for x in th:
        x.join(30)
    assert res["a"] == res["c"] == ["a", "b", "c"]
    for g in (g0, g1, g2):
        g.close()
    with pytest.raises(TypeError):
        _encode({1: "x"})
    with pytest.raises(TypeError):
        _encode(np.arange(3))
    assert _encode({"k": np.float32(1.5), "t": (1, 2)}) == {"k": 1.5, "t": [1, 2]}


def test_host_group_survives_a_stale_rendezvous_file(tmp_path):
    """A launch that died leaves its file behind; the next one with the same key must not trip over the dead port in it."""
    import threading
    import time
    from transflow_amd.batch import HostGroup
    path = str(tmp_path / "rdzv")
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        dead = s.getsockname()[1]
    fd = os.open(path, os.O_WRONLY | os.O_CREAT | os.O_EXCL, 0o600)
    with os.fdopen(fd, "w") as f:
        f.write(f"127.0.0.1:{dead} {'0' * 32} {'1' * 32}\n")
    box = {}
    t1 = threading.Thread(target=lambda: box.setdefault("g1", HostGroup(1, 2, path=path, timeout=60)))
    t1.start()                       # rank 1 first: it finds the stale file and keeps trying
    time.sleep(0.5)
    g0 = HostGroup(0, 2, path=path, timeout=60)
    t1.join(60)
    g1 = box["g1"]
    th = threading.Thread(target=lambda: box.setdefault("r0", g0.max_over_ranks(1.0)))
    th.start()
    assert g1.max_over_ranks(3.0) == 3.0
    th.join(30)
    assert box["r0"] == 3.0
    g0.close()
    g1.close()


NO_RCCL_WORKER = """
import json, os, sys, types
sys.path.insert(0, %r)
import numpy as np
import bench
from transflow_amd import batch as B
import transflow_amd.device as D

rank, _, world = B.env_world()

class FakeGroup:                      # stands in for B.RcclGroup: comes up on rank 0, raises on rank 1
    closed = False
    def __init__(self, host):
        if host.rank == 1:
            raise RuntimeError("stub: ncclCommInitRank refused (two ranks on one device)")
    def broadcast_dev(self, ptr, n):
        pass
    def close(self):
        FakeGroup.closed = True
    def abandon(self):
        FakeGroup.closed = True

class FakeBuffer:                     # stands in for transflow_amd.device.DevBuffer: no GPU in this test
    ptr = 4096
    def __init__(self, n):
        self.n = n
    def upload(self, a):
        self.a = a
    def download(self, shape, dtype):
        return np.zeros(shape, dtype)
    def close(self):
        pass

B.RcclGroup = FakeGroup
D.DevBuffer = FakeBuffer
lib = types.SimpleNamespace(tf_sync=lambda: 0)
host = B.HostGroup(rank, world)
wl = dict(bench.WORKLOADS["4k"], w=64, h=48)
got = bench.rccl_or_nothing(host, rank, wl, lib, lambda rc: None, timeout=30)
res = {"rank": rank, "all_none": all(v is None for v in got[:4]), "error": got[4], "closed": FakeGroup.closed,
       "code": bench.missing_rccl_exit_code(world, False, 0 if got[0] is None else world, False),
       "code_allowed": bench.missing_rccl_exit_code(world, False, 0 if got[0] is None else world, True)}
print("RESULT " + json.dumps(res), flush=True)
host.close()
"""


def test_a_communicator_that_fails_on_one_rank_is_dropped_on_all_and_costs_the_exit_code(tmp_path):
    """SURVEY 8e / BASELINE configs[4]: `bench.py --gpus N` whose RCCL communicator does not come up still measures
    (the path needs no collective) and prints its line, but it is not the run that was asked for: all or nothing --
    the rank whose communicator DID come up closes it -- and exit code 6 unless --allow-no-rccl.  RcclGroup and the
    device buffer are stand-ins here (no GPU); the real thing runs in tests/test_gpu_batch.py."""
    script = tmp_path / "worker.py"
    script.write_text(NO_RCCL_WORKER % ROOT)
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE="2", TF_BATCH_RDZV=str(tmp_path / "rdzv"))
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.PIPE, text=True))
    results = {}
    for p in procs:
        out, err = p.communicate(timeout=180)
        assert p.returncode == 0, err[-2000:]
        r = json.loads([l for l in out.splitlines() if l.startswith("RESULT ")][0][7:])
        results[r["rank"]] = r
    for r in (0, 1):
        assert results[r]["all_none"] and "ncclCommInitRank refused" in results[r]["error"]
        assert results[r]["code"] == 6 and results[r]["code_allowed"] == 0
    assert results[0]["closed"] and not results[1]["closed"]      # rank 0 had a communicator and gave it up


def test_missing_rccl_exit_code_rules():
    sys.path.insert(0, ROOT)
    import bench
    f = bench.missing_rccl_exit_code
    assert f(1, False, 0, False) == 0                       # one rank, no --rccl: nothing was asked for
    assert f(1, True, 1, False) == 0 and f(1, True, 0, False) == 6 and f(1, True, 0, True) == 0
    assert f(8, False, 8, False) == 0 and f(8, False, 0, False) == 6 and f(8, False, 0, True) == 0
    assert bench.EXIT_NO_RCCL == 6
