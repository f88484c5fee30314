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
        with HipFlowSource.from_args(ArrayFrameProvider(frames, 25.0), direction="backward", cv_config=cfg) as source:
            meta.put((source.width, source.height, source.framerate, source.length))
            for flow in source:
                queue.put(flow)
        queue.put(None)
    except Exception as err:  # surfaces in the parent instead of hanging it
        queue.put(err)


def test_forked_flow_process_and_main_compositor():
    """The reference's process layout: the flow source lives in a forked child (its own HIP
    context, created after the fork), flows cross a multiprocessing.Queue(maxsize=1) as pickled
    numpy arrays (pipeline.py:326-328), the compositor runs in the parent.
    Then the same with FlowConfig.hip_device_flows = "ipc": what crosses the queue is a 64-byte HIP IPC handle per flow
    (transflow_amd/deviceflow.py), the parent's compositor reads the flows in HBM -- same frames bit for bit, same flows
    when brought down, and a pickled checkpoint of such a flow holds the host array, never a device address.  Both
    children are forked before this process touches the GPU."""
    import multiprocessing as mp
    import pickle

    from transflow_amd import _lib
    if _lib.load().tf_is_initialized():
        pytest.skip("this process already initialised HIP; a forked child could not use the GPU")
    from transflow_amd.compositor import HipCompositor
    from transflow_amd.config import LayerConfig
    from transflow_amd.flow import ArrayFrameProvider, HipFlowSource
    h, w = 96, 128
    frames = _frames(h, w, 5, seed=33)
    ctx = mp.get_context("fork")
    queue, meta = ctx.Queue(maxsize=1), ctx.Queue()
    child = ctx.Process(target=_child_flow_process, args=(frames, queue, meta))
    child.start()            # forked BEFORE this process touches the GPU in this test's objects
    queue2, meta2 = ctx.Queue(maxsize=1), ctx.Queue()
    child2 = ctx.Process(target=_child_flow_process, args=(frames, queue2, meta2, "ipc"))
    child2.start()           # (it fills its queue and waits there until the first run is over)
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
