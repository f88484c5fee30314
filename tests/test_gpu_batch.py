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
    # the gather beside the library stream's next work: begin -> (other work) -> end, one at a time
    out2 = DevBuffer(a.nbytes)
    g.gather_begin(buf.ptr, a.nbytes, out2.ptr, [a.nbytes])
    with pytest.raises(ValueError):
        g.gather_begin(buf.ptr, a.nbytes, out2.ptr, [a.nbytes])    # the previous one has not been ended
    g.gather_end()
    g.gather_end()                                                 # nothing pending: a no-op
    _lib.check(_lib.load().tf_sync())
    np.testing.assert_array_equal(out2.download(a.shape, np.uint8), a)
    # the gather with a place of its own for the bytes (flows to root: a pass's flows at their clip position)
    out3 = DevBuffer.from_array(np.zeros(a.nbytes + 4096, np.uint8))
    g.gather_at(buf.ptr, a.nbytes, out3.ptr, [a.nbytes], [4096], a.nbytes + 4096)
    _lib.check(_lib.load().tf_sync())
    got = out3.download((a.nbytes + 4096,), np.uint8)
    assert not got[:4096].any()
    np.testing.assert_array_equal(got[4096:], a)
    g.gather_at(0, 0, out3.ptr, [0], [0], a.nbytes + 4096)        # a rank that has run out of passes
    with pytest.raises(ValueError):
        g.gather_at(buf.ptr, a.nbytes, out3.ptr, [a.nbytes], [4097], a.nbytes + 4096)   # leaves the buffer
    with pytest.raises(ValueError):
        g.gather_at(buf.ptr, a.nbytes, out3.ptr, [a.nbytes - 1], [0], a.nbytes + 4096)  # root's own count
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
                          "40", "--batch", "7", "--steps", "3", "--warmup", "1", "--no-extra", "--no-cpu-baseline"],
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-3000:]
    d = json.loads(out.stdout.strip().splitlines()[-1])
    assert d["n_gpus"] == 1 and d["rccl_ranks"] == 1 and "rccl_error" not in d and "side_leg_errors" not in d
    assert d["parity_gate"]["ok"] and d["parity_gate"]["remap_bit_exact"]
    assert d["gather"]["verified_crc"] is True and d["gather"]["frames_per_gather"] == 7
    assert d["gather"]["frames_per_rank"] == [7]
    assert d["gather"]["frames_per_s_with_gather_beside_the_next_step"] > 0
    assert d["config"]["pairs_per_rank"] == [39] and d["config"]["frame_pairs_per_step_per_gpu"] == [7]
    assert d["value"] > 0
    # flows to root (SURVEY 8e mode F): six passes of seven pairs (the last moved back to end with the clip) land at
    # their clip positions; the root's one stream paints what rank 0 paints from the same flows
    f = d["flows_to_root"]
    assert f["clip_pairs"] == 39 and f["gathers"] == 6 and f["bytes_into_root"] == 0
    assert f["verified_flow_crc"] is True and f["verified_stream"] is True and f["frames_per_s"] > 0


def test_bench_line_carries_the_round_6_keys_at_a_small_size():
    """bench.py with its CPU baseline at a small frame size: the gate on three pairs (first, middle, last), the re-check's
    oracle statement, `parity_wide` (the multi-core sample's oracle flows against the GPU's flows of the same pass), the
    untimed steps, the top-level verdict keys of the multi-GPU legs, and no counter figure where profiles/ holds no
    table for the workload."""
    env = dict(os.environ, TF_BENCH_CPU_THREADS="5")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--size", "640x360", "--clip-frames", "17", "--batch",
                          "8", "--steps", "3", "--warmup", "2", "--burn-in", "4", "--no-extra"],
                         env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-3000:]
    d = json.loads(out.stdout.strip().splitlines()[-1])
    g = d["parity_gate"]
    assert g["ok"] and g["pairs"] == [0, 4, 7] and g["pairs_checked"] == 3 and g["exact_pairs"] == [0, 7]
    assert g["outliers_default"] == 0 and g["exact_bit_identical"] and g["remap_bit_exact"]
    r = d["timed_region_recheck"]["per_rank"][0]
    assert r["ok"] and r["pairs"] == [0, 4, 7]
    if r["pass"] == 0:
        assert r["equals_gate_flows"] and r["vs_oracle"]["outliers"] == 0 and r["vs_oracle"]["flow_tol"] > 0
    w = d["parity_wide"]
    assert w == d["cpu_baseline"]["parity_of_the_multi_core_sample"]
    assert w["ok"] and w["pairs"] == 5 and w["outliers"] == 0 and w["pairs_bit_identical"] + (w["pixels_differing"] > 0) >= 1
    assert d["untimed_steps_before_timed_region"] == 6 and d["burn_in_steps"] == 4 and d["warmup"] == 2
    assert d["rccl_ranks"] == 0 and d["gather_verified_crc"] is None and d["flows_to_root_ok"] is None
    ws = d["roofline"]["whole_step"]
    assert ws["counter_GBs"] is None and ws["counter_frac"] is None and "no counter table" in ws["counter_source"]
    assert d["cpu_baseline"]["cores"] == 1 and d["cpu_baseline"]["multi_core"]["cores"] == 5


def test_flows_to_root_paints_the_clip_as_one_compositor_would():
    """SURVEY 8e mode F end to end on a communicator of one rank: the passes' flows gathered to their clip positions
    (the last pass repeats a pair of the one before), ONE remap recurrence over the clip in order on the root; layer
    state, rgba and every frame bit-exact against the oracle's moveref layer fed the same flows in the same order --
    what the reference's one compositor paints for the clip (pipeline.py:565)."""
    sys.path.insert(0, ROOT)
    import bench
    from oracle import remap_ref as OR
    from transflow_amd.batch import HostGroup, RcclGroup, flows_to_root_calls
    from transflow_amd.device import DevBuffer
    from transflow_amd.remap import CompImage
    wl = dict(bench.WORKLOADS["4k"], w=208, h=120)
    w, h = wl["w"], wl["h"]
    plan = bench.make_plan(12, 4, 0, 1)
    assert plan["pass_starts"] == [0, 4, 7]
    job = bench.Job(wl, 4, plan, 12, 77, 0, lanes=1)
    host = HostGroup(0, 1)
    g = RcclGroup(host)
    flow_bytes = w * h * 8
    calls = flows_to_root_calls([plan], flow_bytes)
    clip = DevBuffer(11 * flow_bytes)
    for k, call in enumerate(calls):
        me = call[0]
        job.calc_pass(k)
        g.gather_at(job.fb.flow_ptr(0), me["send_bytes"], clip.ptr, me["recv_bytes"], me["recv_offsets"], me["recv_capacity"])
    job.sync()
    flows = clip.download((11, h, w, 2), np.float32)
    for k, s0 in enumerate(plan["pass_starts"]):               # every pass sits at its clip position
        job.calc_pass(k)
        job.sync()
        for i in range(4):
            np.testing.assert_array_equal(job.fb.get_flow(i), flows[s0 + i])
    layer = job.make_layer()
    comp = CompImage(h, w, (255, 255, 255))
    ora = OR.MoveRefLayer(h, w, OR.LayerParams(reset_mode="random", reset_random_factor=0.5), reset_mask=job.reset_mask,
                          introduction_masks=[np.ones((h, w), bool)])
    white = np.full((h, w, 3), 255, np.uint8)
    ubuf = DevBuffer(h * w * 8)
    for j in range(11):
        layer.uniform_dev(bench.SEED_U, ubuf.ptr)
        u = ubuf.download((h, w), np.float64)
        layer.step_dev(comp, clip.ptr + j * flow_bytes, job.pixmap_dev, 3, clip_flow=True, seed=bench.SEED_U)
        ora.update(OR.post_process(flows[j].copy(), wl["direction"]), [job.pixmap], u)
        np.testing.assert_array_equal(comp.download(), OR.composite(white, [ora.render()]))
    data, rgba = layer.get_state()
    np.testing.assert_array_equal(data, ora.data)
    np.testing.assert_array_equal(rgba, ora.rgba)
    g.close()
    host.close()


def test_compositor_image_in_the_callers_buffer():
    """tf_comp_create_on: the frames of a batch side by side in one device buffer (what one gather sends)."""
    from transflow_amd.device import DevBuffer
    from transflow_amd.remap import CompImage, RemapLayer
    h, w = 37, 53
    rng = np.random.default_rng(3)
    buf = DevBuffer(3 * h * w * 3)
    comps = [CompImage(h, w, (1, 2, 3), image_dev=buf.ptr + i * h * w * 3) for i in range(3)]
    own = CompImage(h, w, (1, 2, 3))
    pix = [rng.integers(0, 256, (h, w, 3), dtype=np.uint8) for _ in range(3)]
    for c, p in zip(comps, pix):
        layer = RemapLayer(h, w)
        layer.set_sources([np.ones((h, w), np.uint8)])
        flow = rng.integers(-2, 3, (h, w, 2)).astype(np.float32)
        from oracle import remap_ref as OR
        flow = OR.post_process(flow, OR.BACKWARD)
        for target in (c, own):
            lay = RemapLayer(h, w)
            lay.set_sources([np.ones((h, w), np.uint8)])
            lay.update(flow)
            lay.gather(0, p)
            target.begin()
            lay.render(target)
        np.testing.assert_array_equal(c.download(), own.download())
    whole = buf.download((3, h, w, 3), np.uint8)
    for i, c in enumerate(comps):
        np.testing.assert_array_equal(whole[i], c.download())
    for c in comps:
        c.close()
    np.testing.assert_array_equal(buf.download((3, h, w, 3), np.uint8), whole)   # the buffer outlives its handles


def test_bench_with_three_ranks_sharing_the_one_gpu():
    """A rehearsal of the multi-rank run on the one GPU of the test box: `bench.py --gpus 3` starts three rank
    processes that share device 0 (TF_BENCH_SHARE_GPU), meet through the host group, shard a 40-frame clip into
    13 + 13 + 13 pairs, time the same steps and report one line.  RCCL refuses a communicator with two ranks on one
    device, so the line says rccl_ranks 0 and why -- every rank then makes the shared inputs itself; the multi-GPU
    data legs are the 8-GPU node's to exercise (tests/test_batch_gloo.py replays their call sequence)."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "TF_BATCH_RDZV")}
    env["TF_BENCH_SHARE_GPU"] = "1"
    env["TF_BENCH_RCCL_TIMEOUT"] = "60"
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "3", "--size", "640x360", "--clip-frames",
                          "40", "--batch", "7", "--steps", "3", "--warmup", "1", "--no-extra", "--no-cpu-baseline"],
                         env=env, capture_output=True, text=True, timeout=900)
    lines = [ln for ln in out.stdout.strip().splitlines() if ln.startswith("{")]
    assert lines, out.stderr[-3000:]
    d = json.loads(lines[-1])
    # round 6: a multi-rank run without its communicator still prints the line, and leaves with exit code 6 so that the
    # fallback cannot be mistaken for the result (--allow-no-rccl, the rehearsal's flag, makes it 0: second run below)
    assert out.returncode == (6 if d["rccl_ranks"] == 0 else 0), (out.returncode, out.stderr[-2000:])
    assert d["gather_verified_crc"] is None and d["flows_to_root_ok"] is None or d["rccl_ranks"] == 3
    assert d["untimed_steps_before_timed_region"] == d["burn_in_steps"] + 1
    if d["rccl_ranks"] == 0:
        assert "exit code 6" in out.stderr
        again = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "3", "--size", "640x360",
                                "--clip-frames", "40", "--batch", "7", "--steps", "2", "--warmup", "1", "--no-extra",
                                "--no-cpu-baseline", "--no-gate", "--burn-in", "0", "--allow-no-rccl"],
                               env=env, capture_output=True, text=True, timeout=900)
        assert again.returncode == 0, again.stderr[-2000:]
    assert d["n_gpus"] == 3 and d["config"]["pairs_per_rank"] == [13, 13, 13]
    assert d["config"]["frame_pairs_per_step_per_gpu"] == [7, 7, 7] and d["config"]["frame_pairs_per_step"] == 21
    assert len(d["per_rank_frames_per_s"]) == 3 and all(v > 0 for v in d["per_rank_frames_per_s"])
    assert d["parity_gate"]["ok"] and d["value"] > 0
    assert d["rccl_ranks"] in (0, 3)
    if d["rccl_ranks"] == 0:
        assert d.get("rccl_error")


@pytest.mark.parametrize("fault,rc,needle", [("gather-raises", 0, "injected failure"), ("gather-hangs", 4, "TimeoutError")])
def test_a_lost_side_leg_never_loses_the_line(fault, rc, needle):
    """The result line is complete before any untimed leg starts: a leg that raises is reported under side_leg_errors
    (exit code 0), one that never returns is abandoned after its time limit, the line is printed all the same and the
    process leaves with exit code 4."""
    env = dict(os.environ, TF_BENCH_LEG_TIMEOUT="5")
    # bench.py with its gather leg replaced (bench.GATHER_LEG): the fault lives here, not in the benchmark
    driver = ("import sys, time; sys.path.insert(0, %r); import bench\n"
              "def faulty(*a, **k):\n"
              "    if %r == 'gather-raises':\n"
              "        raise RuntimeError('injected failure of the gather leg')\n"
              "    time.sleep(3600)\n"
              "bench.GATHER_LEG = faulty\n"
              "sys.argv = ['bench.py'] + sys.argv[1:]\n"
              "bench.run_as_main()\n") % (ROOT, fault)
    out = subprocess.run([sys.executable, "-c", driver, "--rccl", "--size", "640x360", "--clip-frames",
                          "9", "--batch", "4", "--steps", "2", "--warmup", "1", "--no-extra", "--no-cpu-baseline"],
                         env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == rc, out.stderr[-2000:]
    d = json.loads([ln for ln in out.stdout.strip().splitlines() if ln.startswith("{")][-1])
    assert d["value"] > 0 and d["parity_gate"]["ok"] and "gather" not in d
    assert needle in d["side_leg_errors"]["gather"]


def test_bench_on_two_gpus_with_real_rccl_when_the_box_has_them():
    """`bench.py --gpus 2` end to end with a communicator of two ranks on two devices: broadcast of the shared inputs,
    sharded clip, barriers, the per-pass gather with its CRC check, one JSON line.  The test box has one GPU (RCCL
    refuses two ranks on one device): skipped there, run wherever tf_device_count reports two."""
    from transflow_amd import _lib
    n = C.c_int()
    _lib.check(_lib.load().tf_device_count(C.byref(n)))
    if n.value < 2:
        pytest.skip(f"{n.value} GPU visible: RCCL needs one device per rank")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--size", "640x360", "--clip-frames", "41",
                          "--batch", "20", "--steps", "3", "--warmup", "1", "--no-extra", "--no-cpu-baseline"],
                         capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    d = json.loads(out.stdout.strip().splitlines()[-1])
    assert d["n_gpus"] == 2 and d["rccl_ranks"] == 2 and "rccl_error" not in d and "side_leg_errors" not in d
    assert d["parity_gate"]["ok"]
    assert d["gather"]["verified_crc"] is True and d["gather"]["frames_per_rank"] == [20, 20]
    f = d["flows_to_root"]
    assert f["clip_pairs"] == 40 and f["gathers"] == 1 and f["bytes_into_root"] == 20 * 640 * 360 * 8
    assert f["verified_flow_crc"] is True and f["verified_stream"] is True
