"""The reference's process layout on the GPU: the flow source lives in a FORKED child process
(pipeline.py:56-101), the compositor in the parent.  HIP cannot be used in a child forked after
the parent initialised it, so this file sorts first among the GPU tests and skips itself if the
process has already touched the GPU."""
import numpy as np
import pytest

from tests.test_gpu_dropin import FakeSource, _frames

pytestmark = pytest.mark.gpu


def _child_flow_process(frames, queue, meta, device_flows=None):
    """What pipeline.py's SourceProcess.run does (pipeline.py:71-101): build the source inside the
    child, report its geometry, then stream flows through a bounded queue."""
    from transflow_amd.config import FlowConfig
    from transflow_amd.flow import ArrayFrameProvider, HipFlowSource
    try:
        cfg = FlowConfig(hip_device_flows=device_flows) if device_flows else None
        ring_report = None
        with HipFlowSource.from_args(ArrayFrameProvider(frames, 25.0), direction="backward", cv_config=cfg) as source:
            meta.put((source.width, source.height, source.framerate, source.length))
            for flow in source:
                queue.put(flow)
            del flow
            ring = getattr(source, "_flow_ring", None)
            if ring is not None:
                # what close() is about to do: wait until every token has been made and acknowledged by the consumer
                source.prev_flow = None
                ring_report = (bool(ring.drain(timeout=120.0)), len(ring._all), ring.exports, ring.unacknowledged())
        if ring_report is not None:
            meta.put(ring_report)
        queue.put(None)
    except Exception as err:  # surfaces in the parent instead of hanging it
        queue.put(err)


@pytest.fixture(scope="module")
def forked():
    """Every child process of this file, forked BEFORE this process touches the GPU (HIP cannot be used in a child
    forked after the parent initialised it): the two flow processes of the first test and the two hand-made producers
    of the second.  They fill their queues and wait there until their test reads them."""
    import multiprocessing as mp

    from transflow_amd import _lib
    if _lib.load().tf_is_initialized():
        pytest.skip("this process already initialised HIP; a forked child could not use the GPU")
    h, w = 96, 128
    frames = _frames(h, w, 5, seed=33)
    ctx = mp.get_context("fork")
    out = {"frames": frames, "h": h, "w": w}
    out["queue"], out["meta"] = ctx.Queue(maxsize=1), ctx.Queue()
    out["child"] = ctx.Process(target=_child_flow_process, args=(frames, out["queue"], out["meta"]))
    out["queue2"], out["meta2"] = ctx.Queue(maxsize=1), ctx.Queue()
    out["child2"] = ctx.Process(target=_child_flow_process, args=(frames, out["queue2"], out["meta2"], "ipc"))
    for tag, overrun in (("bad", True), ("good", False)):
        q, r, go = ctx.Queue(), ctx.Queue(), ctx.Event()
        out[tag] = (ctx.Process(target=_child_ring_producer, args=(q, r, go, overrun)), q, r, go)
    for proc in (out["child"], out["child2"], out["bad"][0], out["good"][0]):
        proc.start()
    yield out
    for proc in (out["child"], out["child2"], out["bad"][0], out["good"][0]):
        if proc.is_alive():
            proc.terminate()


def test_forked_flow_process_and_main_compositor(forked):
    """The reference's process layout: the flow source lives in a forked child (its own HIP
    context, created after the fork), flows cross a multiprocessing.Queue(maxsize=1) as pickled
    numpy arrays (pipeline.py:326-328), the compositor runs in the parent.
    Then the same with FlowConfig.hip_device_flows = "ipc": what crosses the queue is a 64-byte HIP IPC handle per flow
    (transflow_amd/deviceflow.py), the parent's compositor reads the flows in HBM -- same frames bit for bit, same flows
    when brought down, and a pickled checkpoint of such a flow holds the host array, never a device address.  Both
    children are forked before this process touches the GPU."""
    import pickle

    from transflow_amd.compositor import HipCompositor
    from transflow_amd.config import LayerConfig
    from transflow_amd.flow import ArrayFrameProvider, HipFlowSource
    h, w, frames = forked["h"], forked["w"], forked["frames"]
    queue, meta, child = forked["queue"], forked["meta"], forked["child"]
    queue2, meta2, child2 = forked["queue2"], forked["meta2"], forked["child2"]   # (fills its queue and waits there)
    assert meta.get(timeout=120) == (w, h, 25.0, 4)
    pixmap = np.random.default_rng(4).integers(0, 256, (h, w, 3), dtype=np.uint8)
    comp = HipCompositor.from_args(h, w, [LayerConfig(0)], "#000000")
    comp.set_sources({0: [FakeSource([pixmap], np.ones((h, w), bool))]})
    got = []
    while True:
        item = queue.get(timeout=120)
        if item is None:
            break
        if isinstance(item, Exception):
            raise item
        comp.update(item)
        got.append((item, comp.render()))
    child.join(timeout=60)
    assert child.exitcode == 0 and len(got) == 4
    # ... and with the flows crossing as IPC handles
    from transflow_amd.deviceflow import DeviceFlow
    assert meta2.get(timeout=120) == (w, h, 25.0, 4)
    comp3 = HipCompositor.from_args(h, w, [LayerConfig(0)], "#000000")
    comp3.set_sources({0: [FakeSource([pixmap], np.ones((h, w), bool))]})
    got_ipc = []
    while True:
        item = queue2.get(timeout=120)
        if item is None:
            break
        if isinstance(item, Exception):
            raise item
        assert isinstance(item, DeviceFlow) and item._host is None      # a token crossed, nothing was brought down
        comp3.update(item)
        got_ipc.append((item, comp3.render()))
    # the producer's side of the protocol: all four tokens were acknowledged before it let go of its buffers, and its
    # ring stayed small (a buffer returns to rotation as soon as its token has been copied)
    drained, n_buffers, exports, pending = meta2.get(timeout=120)
    assert drained and exports == 4 and pending == [] and n_buffers <= 6, (drained, n_buffers, exports, pending)
    child2.join(timeout=60)
    assert child2.exitcode == 0 and len(got_ipc) == 4
    for (flow_host, frame_host), (flow_dev, frame_dev) in zip(got, got_ipc):
        np.testing.assert_array_equal(frame_host, frame_dev)
        assert flow_dev._host is None
        blob = pickle.dumps({"flow": flow_dev})                          # a checkpoint: the host array only
        back = pickle.loads(blob)["flow"]
        assert type(back) is np.ndarray
        np.testing.assert_array_equal(back, flow_host)
        np.testing.assert_array_equal(np.asarray(flow_dev), flow_host)
    # the same run in one process gives the same flows and frames, bit for bit
    comp2 = HipCompositor.from_args(h, w, [LayerConfig(0)], "#000000")
    comp2.set_sources({0: [FakeSource([pixmap], np.ones((h, w), bool))]})
    with HipFlowSource.from_args(ArrayFrameProvider(frames, 25.0), direction="backward") as source:
        for (flow_child, frame_child), flow in zip(got, source):
            np.testing.assert_array_equal(flow_child, flow)
            comp2.update(flow)
            np.testing.assert_array_equal(frame_child, comp2.render())


def _child_ring_producer(queue, report, go, overrun):
    """A producer process that makes flows by hand in a FlowRing of two buffers and sends them as IPC tokens.
    overrun = True: the ring is told not to wait for acknowledgements (the test switch) and writes buffer 0 again while
    its first token is still in the queue.  overrun = False: the ordinary ring; the producer sends its last flow, drains
    and ENDS while the consumer has not looked at the queue yet."""
    import ctypes as C
    import time

    from transflow_amd import _lib, deviceflow as DF
    try:
        lib = _lib.load()
        _lib.check(lib.tf_init(0))
        shape = (24, 40, 2)
        ring = DF.FlowRing(shape, slots=2, wait_for_acks=not overrun)

        def make(value):
            slot = ring.take()
            a = np.full(shape, value, np.float32)
            _lib.check(lib.tf_dev_upload(C.c_void_p(slot.flow_ptr), C.c_void_p(a.ctypes.data), a.nbytes))
            slot.ready.record()
            return DF.DeviceFlow(shape, slot.flow_ptr, slot.ready, ring=ring, slot=slot, cross_process="ipc"), slot.index

        indices = []
        flow, i = make(1.0)
        indices.append(i)
        queue.put(flow)
        del flow
        t0 = time.monotonic()
        while ring.exports < 1 and time.monotonic() - t0 < 30:       # the feeder thread has made the token
            time.sleep(0.001)
        for v in (2.0, 3.0):                                           # more flows while the first token waits in the queue
            flow, i = make(v)
            indices.append(i)
            _lib.check(lib.tf_sync())
            del flow
        report.put(("indices", indices, ring.unacknowledged()))
        if overrun:
            go.wait(900)                                               # the consumer reads the (stale) token now
            report.put(("drained", ring.drain(timeout=20.0)))
        else:
            flow, i = make(4.0)
            queue.put(flow)
            del flow
            report.put(("drained", ring.drain(timeout=900.0)))         # returns only once BOTH tokens were copied
        queue.put(None)
    except Exception as err:      # surfaces in the parent instead of hanging it
        report.put(("error", repr(err)))
        queue.put(err)


def test_ipc_tokens_are_acknowledged_and_an_overrun_raises_instead_of_passing_a_wrong_flow(forked):
    """transflow_amd/deviceflow.py's two guards for a flow that crosses the reference's queue (pipeline.py:85-86, 326) as
    an IPC token.  (1) A producer that overruns its consumer -- forced here with the ring's test switch: buffer 0 is
    written again while its first token still waits in the queue -- makes the consumer's queue.get() raise a RuntimeError
    that names the buffer, not hand out the newer flow under the older one's name.  (2) The ordinary ring never does
    that: with the first token unacknowledged it takes other buffers, and its drain() -- what HipFlowSource.close() runs
    before the producer lets go -- returns only after the consumer, which looks at the queue a second late, has copied
    both flows out intact (multiprocessing's Queue.get() frees the queue's slot before it unpickles, so without the
    wait the producer could end first)."""
    import time

    from transflow_amd.deviceflow import DeviceFlow
    bad, q1, r1, go1 = forked["bad"]
    good, q2, r2, go2 = forked["good"]
    # (1) the overrun
    kind, indices, pending = r1.get(timeout=120)
    assert kind == "indices", indices
    assert indices[0] == 0 and 0 in indices[1:], indices          # buffer 0 was handed out again, unacknowledged
    with pytest.raises(RuntimeError, match=r"device flow buffer 0 of process \d+ was written again"):
        q1.get(timeout=120)
    go1.set()
    assert r1.get(timeout=120) == ("drained", True)               # the failed read still acknowledged: nobody waits for it
    assert q1.get(timeout=120) is None
    bad.join(timeout=60)
    assert bad.exitcode == 0
    # (2) the ordinary ring, and a consumer that comes late
    kind, indices, pending = r2.get(timeout=120)
    assert kind == "indices", indices
    assert indices[0] == 0 and 0 not in indices[1:] and pending == [0], (indices, pending)
    time.sleep(0.5)                                               # the producer sent its last flow long ago and sits in drain()
    assert good.is_alive()
    first = q2.get(timeout=120)
    second = q2.get(timeout=120)
    for flow, value in ((first, 1.0), (second, 4.0)):
        assert isinstance(flow, DeviceFlow) and flow._host is None
        a = np.asarray(flow)
        assert a.shape == (24, 40, 2) and float(a.min()) == float(a.max()) == value
    assert r2.get(timeout=120) == ("drained", True)
    assert q2.get(timeout=120) is None
    good.join(timeout=60)
    assert good.exitcode == 0
