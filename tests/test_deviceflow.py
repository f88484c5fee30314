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


def test_a_buffer_that_left_as_an_ipc_token_stays_out_of_rotation(monkeypatch):
    """A flow that crosses a multiprocessing queue as an IPC handle is garbage in the producer as soon as it is pickled;
    the other process reads the buffer later, inside its queue.get().  The ring must not hand that buffer out again until
    EXPORT_HOLD more flows have left the same way (CPU: the device buffers and events are stood in for)."""
    from transflow_amd import deviceflow as DF

    class Slot:
        def __init__(self, nbytes, index):
            self.index, self.used, self.exported_at, self.ipc_handle = index, None, None, None
    monkeypatch.setattr(DF, "_Slot", Slot)
    ring = DF.FlowRing((4, 5, 2), slots=4)
    a = ring.take()
    ring.give_back(a)
    assert ring.take() is a                       # an ordinary buffer: back at once
    ring.exported(a)                              # a flow in it left as a token ...
    ring.give_back(a)                             # ... and was dropped by the producer
    seen = []
    for _ in range(DF.FlowRing.EXPORT_HOLD - 1):
        s = ring.take()
        assert s is not a
        seen.append(s)
        ring.exported(s)
        ring.give_back(s)
    assert ring.take() is a                       # EXPORT_HOLD exports later it is in rotation again
