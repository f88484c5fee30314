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

from transflow_amd.batch import batch_starts, frames_needed, shard_range

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
    pairs = [tuple(p["pairs"]) for p in d["plans"]]
    assert pairs == [shard_range(255, r, world) for r in range(world)]
    for r, p in enumerate(d["plans"]):
        assert p["rank"] == r
        assert tuple(p["frames"]) == (p["pairs"][0], p["pairs"][1] + 1)        # the halo frame
        assert p["pairs_per_pass"] == min(32, p["n_pairs"])                        # bench.py's default --batch
        assert p["pass_starts"] == batch_starts(p["n_pairs"], 32)


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
