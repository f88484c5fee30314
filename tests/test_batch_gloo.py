"""Batch-of-frames mode over several ranks, on CPU: world_size-2 gloo processes run the
same rendezvous / broadcast / barrier / max-over-ranks / gather code bench.py uses with
RCCL, plus the frame sharding arithmetic."""
import os
import socket
import subprocess
import sys
import textwrap

import numpy as np
import pytest

from transflow_amd.batch import frames_needed, shard_range

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("total,world", [(256, 8), (257, 8), (7, 8), (0, 3), (10, 1), (64, 2)])
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


WORKER = textwrap.dedent("""
    import os, sys, json
    import numpy as np
    sys.path.insert(0, %r)
    from transflow_amd.batch import Group, shard_range
    g = Group("gloo")
    assert g.world == 2
    rng = np.random.default_rng(7)
    pix = rng.integers(0, 256, (6, 8, 3), dtype=np.uint8) if g.rank == 0 else np.zeros((6, 8, 3), np.uint8)
    pix = g.broadcast_bytes(pix, src=0)                      # shared pixmap from rank 0
    lo, hi = shard_range(9, g.rank, g.world)                 # this rank's frame pairs
    g.barrier()
    elapsed = g.max_over_ranks(1.0 + g.rank)                 # timing = slowest rank
    total = g.sum_over_ranks(hi - lo)
    frames = g.gather_arrays(np.full((2, 3), g.rank, np.uint8), dst=0)
    out = {"rank": g.rank, "pix_sum": int(pix.sum()), "span": [lo, hi], "elapsed": elapsed, "total": total,
           "gathered": None if frames is None else [int(f[0, 0]) for f in frames]}
    print("RESULT " + json.dumps(out), flush=True)
    g.close()
""")


def test_two_rank_gloo_group(tmp_path):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    script = tmp_path / "worker.py"
    script.write_text(WORKER % ROOT)
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.PIPE, text=True))
    results = {}
    for p in procs:
        out, err = p.communicate(timeout=180)
        assert p.returncode == 0, err[-2000:]
        import json
        line = [l for l in out.splitlines() if l.startswith("RESULT ")][0]
        r = json.loads(line[7:])
        results[r["rank"]] = r
    expected_sum = int(np.random.default_rng(7).integers(0, 256, (6, 8, 3), dtype=np.uint8).sum())
    assert results[0]["pix_sum"] == results[1]["pix_sum"] == expected_sum
    assert results[0]["span"] == [0, 5] and results[1]["span"] == [5, 9]
    assert results[0]["elapsed"] == results[1]["elapsed"] == 2.0
    assert results[0]["total"] == results[1]["total"] == 9
    assert results[0]["gathered"] == [0, 1] and results[1]["gathered"] is None
