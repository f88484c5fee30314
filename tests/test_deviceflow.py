"""transflow_amd.deviceflow.DeviceFlow on the CPU: what it does once its values are on the host (the download itself
needs a GPU: tests/test_gpu_dropin.py).  It must behave as the float32 (H, W, 2) array it stands for wherever the
reference's pipeline touches a flow (pipeline.py:149-158, 502-506, 565), and an ordinary pickle of it must be the pickle
of that array -- a checkpoint never holds a device address."""
import copy
import pickle

import numpy as np
import pytest

from transflow_amd.deviceflow import DeviceFlow


def _flow(h=5, w=7, seed=0):
    a = np.random.default_rng(seed).normal(0, 2, (h, w, 2)).astype(np.float32)
    f = DeviceFlow(a.shape, 0xdead0000, None)
    f._host = a.copy()          # as if it had been brought down
    return f, a


def test_it_is_the_array_it_stands_for():
    f, a = _flow()
    assert f.shape == a.shape and f.dtype == np.float32 and f.ndim == 3 and len(f) == 5 and f.size == a.size
    np.testing.assert_array_equal(np.asarray(f), a)
    np.testing.assert_array_equal(np.ascontiguousarray(f, dtype=np.float64), a.astype(np.float64))
    np.testing.assert_array_equal(f[..., 0], a[..., 0])
    np.testing.assert_array_equal(f * 2 + 1, a * 2 + 1)
    np.testing.assert_array_equal(1 - f, 1 - a)
    np.testing.assert_array_equal(np.round(f).astype(int), np.round(a).astype(int))     # pipeline.py:506
    np.testing.assert_array_equal(np.maximum(f, f), a)
    np.testing.assert_array_equal(np.stack([f, f]).mean(axis=0), a)                     # merging, pipeline.py:149-158
    np.testing.assert_array_equal(f.copy(), a)
    np.testing.assert_array_equal(f.reshape(-1, 2), a.reshape(-1, 2))
    assert float(f.max()) == float(a.max()) and f.tobytes() == a.tobytes()
    # looking does not change anything: readers get read-only views and the device copy stays the current one
    assert not f.on_host and not np.asarray(f).flags.writeable and not f[..., 0].flags.writeable
    with pytest.raises(ValueError):
        np.asarray(f)[0, 0, 0] = 1.0
    # writing goes through the flow: in place, like an array, and the compositor is told to take the host values
    f[0, 0] = (9, 9)
    assert f.on_host and tuple(np.asarray(f)[0, 0]) == (9.0, 9.0)
    f *= 2
    assert isinstance(f, DeviceFlow)
    np.testing.assert_array_equal(np.asarray(f)[1:], a[1:] * 2)
    np.clip(f, -1, 1, out=f)
    assert float(np.abs(f).max()) <= 1.0 and "DeviceFlow" in repr(f)


def test_an_ordinary_pickle_is_the_host_array():
    f, a = _flow(seed=3)
    for back in (pickle.loads(pickle.dumps(f)), copy.deepcopy(f), pickle.loads(pickle.dumps({"flow": f}))["flow"]):
        assert type(back) is np.ndarray and back.dtype == np.float32
        np.testing.assert_array_equal(back, a)
    assert b"dead0000" not in pickle.dumps(f) and str(0xdead0000).encode() not in pickle.dumps(f)


def test_through_a_multiprocessing_pickler_without_ipc_it_is_the_array_too():
    """ForkingPickler is what multiprocessing queues use; without hip_device_flows = "ipc" (or once the flow has been
    brought down) the array crosses, as in the reference."""
    from multiprocessing.reduction import ForkingPickler
    f, a = _flow(seed=4)
    back = pickle.loads(bytes(ForkingPickler.dumps(f)))
    assert type(back) is np.ndarray
    np.testing.assert_array_equal(back, a)
    f._cross = "ipc"            # asked for, but the values are on the host already: still the array
    back = pickle.loads(bytes(ForkingPickler.dumps(f)))
    assert type(back) is np.ndarray


@pytest.fixture()
def cpu_ring(monkeypatch):
    """FlowRing on the CPU: the device buffer, the events and tf_dev_store_u64 are stood in for; the generation words a
    ring stores are recorded in `stores` as (buffer address, generation)."""
    import types

    from transflow_amd import deviceflow as DF
    stores = []

    class Slot:
        count = 0

        def __init__(self, nbytes, index):
            Slot.count += 1
            self.buf = types.SimpleNamespace(ptr=0x1000 * Slot.count)
            self.flow_ptr = self.buf.ptr + DF.HEADER
            self.index, self.used, self.ipc_handle, self.ready = index, None, None, None
            self.generation, self.exported_gen = 0, None

    stub = types.SimpleNamespace(tf_dev_store_u64=lambda p, v: stores.append((p.value, int(v))) or 0)
    monkeypatch.setattr(DF, "_Slot", Slot)
    monkeypatch.setattr(DF, "_lib", types.SimpleNamespace(load=lambda: stub))
    monkeypatch.setattr(DF, "check", lambda rc: None)
    return DF, stores


def test_a_buffer_that_left_as_an_ipc_token_stays_out_of_rotation_until_it_is_acknowledged(cpu_ring):
    """A flow that crosses a multiprocessing queue as an IPC handle is garbage in the producer as soon as it is pickled;
    the other process reads the buffer later, inside its queue.get().  The ring does not hand that buffer out again
    until the consumer has acknowledged the generation that left (rounds 1-5: until eight more exports had happened --
    an assumption about the queue's depth); it grows meanwhile.  Every hand-out starts a new generation, stored on the
    device before anything else is queued for the buffer."""
    DF, stores = cpu_ring
    ring = DF.FlowRing((4, 5, 2), slots=4)
    a = ring.take()
    assert (a.generation, stores) == (1, [(a.buf.ptr, 1)])
    ring.give_back(a)
    assert ring.take() is a and a.generation == 2          # an ordinary buffer: back at once, in a new generation
    gen, board = ring.exported(a)                          # a flow in it leaves as a token ...
    assert gen == 2 and board and ring.unacknowledged() == [a.index]
    ring.give_back(a)                                      # ... and is dropped by the producer
    others = []
    for _ in range(12):                                    # however many flows follow: not this buffer
        s = ring.take()
        assert s is not a
        others.append(s)
    for s in others:
        ring.give_back(s)
    assert not ring.drain(timeout=0.05)                    # the producer may not let go yet
    DF._acknowledge(board, a.index, 1)                     # an older generation's acknowledgement does not count
    assert ring.unacknowledged() == [a.index]
    DF._acknowledge(board, a.index, gen)                   # the consumer has copied it
    assert ring.unacknowledged() == [] and ring.drain(timeout=0.05)
    assert ring.take() is a and a.generation == 3 and stores[-1] == (a.buf.ptr, 3)
    # the test switch: a ring told not to wait hands the buffer out again at once (tests/test_gpu_a_forked_pipeline.py
    # uses it to overrun a consumer on purpose)
    fast = DF.FlowRing((4, 5, 2), slots=2, wait_for_acks=False)
    b = fast.take()
    fast.exported(b)
    fast.give_back(b)
    assert fast.take() is b
    DF._acknowledge("/nonexistent/tfhip-ack", 0, 1)        # a board that is gone is not an error
    DF._acknowledge(board, DF._AckBoard.WORDS + 3, 1)      # nor is a buffer beyond the board


def test_drain_waits_for_flows_on_their_way_into_a_token(cpu_ring):
    """multiprocessing pickles in a feeder thread, after put() has returned: a flow of the "ipc" kind counts as on its
    way from its creation until it has been exported, read on the host, or dropped."""
    DF, _ = cpu_ring
    ring = DF.FlowRing((4, 5, 2), slots=2)
    s = ring.take()
    f = DF.DeviceFlow((4, 5, 2), s.flow_ptr, None, ring=ring, slot=s, cross_process="ipc")
    assert ring._in_transit == 1 and not ring.drain(timeout=0.02)
    f._left_transit()                                      # what the queue's reducer does once the token is made
    assert ring._in_transit == 0
    f._left_transit()                                      # idempotent
    assert ring._in_transit == 0 and ring.drain(timeout=0.02)
    g = DF.DeviceFlow((4, 5, 2), s.flow_ptr, None, ring=ring, slot=s, cross_process="ipc")
    assert ring._in_transit == 1
    del g                                                  # a flow nobody sent anywhere
    assert ring._in_transit == 0
    h = DF.DeviceFlow((4, 5, 2), s.flow_ptr, None, ring=ring, slot=s)      # in-process flows never count
    assert ring._in_transit == 0
    del h, f
