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
