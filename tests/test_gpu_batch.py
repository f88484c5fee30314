"""The RCCL legs of the batch-of-frames mode (tf_batch_*) on the one GPU of the test
box: a communicator of one rank still goes through ncclGetUniqueId / ncclCommInitRank / ncclBroadcast /
send-recv gather / ncclAllReduce, through the C ABI and without torch.  More ranks need more GPUs (RCCL
refuses two ranks on one device): the driver's 8-GPU run covers them, the CPU tests cover the launcher."""
import ctypes as C
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_rccl_group_of_one_rank_through_the_c_abi():
    from transflow_amd import _lib
    from transflow_amd.batch import HostGroup, RcclGroup
    from transflow_amd.device import DevBuffer
    assert "torch" not in sys.modules
    host = HostGroup(0, 1)
    g = RcclGroup(host)
    assert g.rank == 0 and g.world == 1 and g.rccl_version > 0
    rng = np.random.default_rng(5)
    a = rng.integers(0, 256, 1 << 20, dtype=np.uint8)
    buf = DevBuffer.from_array(a)
    g.broadcast_dev(buf.ptr, a.nbytes)
    np.testing.assert_array_equal(buf.download(a.shape, np.uint8), a)
    out = DevBuffer(a.nbytes + 64)
    g.gather_dev(buf.ptr, a.nbytes, out.ptr, [a.nbytes])
    _lib.check(_lib.load().tf_sync())
    np.testing.assert_array_equal(out.download(a.shape, np.uint8), a)
    g.gather_dev(buf.ptr, a.nbytes, out.ptr)                 # equal counts form
    assert g.reduce([1.5, -2.0], "max") == [1.5, -2.0]
    assert g.reduce([1.5, -2.0], "sum") == [1.5, -2.0]
    g.barrier()
    with pytest.raises(ValueError):
        g.gather_dev(buf.ptr, a.nbytes, out.ptr, [a.nbytes + 1])   # root's own count must match what it sends
    with pytest.raises(ValueError):
        g.broadcast_dev(buf.ptr, 16, root=3)
    g.close()
    host.close()
    assert "torch" not in sys.modules


def test_bench_runs_its_rccl_legs_with_one_rank():
    """bench.py --rccl at a small frame size: shared inputs through tf_batch_broadcast, the gather leg
    verified by CRC, the parity gate on, one JSON line out."""
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--rccl", "--size", "640x360", "--clip-frames",
                          "9", "--batch", "4", "--steps", "3", "--warmup", "1", "--no-extra", "--no-cpu-baseline"],
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-3000:]
    d = json.loads(out.stdout.strip().splitlines()[-1])
    assert d["n_gpus"] == 1 and d["rccl_ranks"] == 1 and "rccl_error" not in d
    assert d["parity_gate"]["ok"] and d["parity_gate"]["remap_bit_exact"]
    assert d["gather"]["verified_crc"] is True
    assert d["config"]["pairs_per_rank"] == [8]
    assert d["value"] > 0
