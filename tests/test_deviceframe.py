"""transflow_amd.deviceframe.DeviceFrame on the CPU: what it is once its transfer has been waited for (the transfer
needs a GPU: tests/test_gpu_dropin.py).  To the reference's pipeline a rendered frame is a uint8 (H, W, 3) array that is
put on the outputs' queue (pipeline.py:518-522) and whose bytes are written to the encoder (output/ffmpeg.py:32-54)."""
import copy
import pickle

import numpy as np

from transflow_amd.deviceframe import DeviceFrame


class _Image:
    """stands in for the CompImage whose download was begun"""

    def __init__(self):
        self.ended = 0

    def download_end(self):
        self.ended += 1


def _frame(h=4, w=6, seed=0):
    a = np.random.default_rng(seed).integers(0, 256, (h, w, 3), dtype=np.uint8)
    img = _Image()
    return DeviceFrame(a.copy(), img), a, img


def test_first_use_waits_for_the_transfer_once():
    f, a, img = _frame()
    assert not f.arrived and img.ended == 0 and "on its way down" in repr(f)
    assert f.shape == a.shape and f.dtype == np.uint8 and f.ndim == 3 and len(f) == 4 and f.size == a.size == f.nbytes
    assert img.ended == 0                                   # shape, dtype, len: no wait
    np.testing.assert_array_equal(np.asarray(f), a)
    assert f.arrived and img.ended == 1
    np.testing.assert_array_equal(f[1:, :2], a[1:, :2])
    assert f.tobytes() == a.tobytes() and f.copy().flags.writeable and img.ended == 1      # output/ffmpeg.py writes the bytes
    np.testing.assert_array_equal(f[..., ::-1], a[..., ::-1])                              # RGB -> BGR for a cv window
    np.testing.assert_array_equal(np.concatenate([f, f]), np.concatenate([a, a]))
    np.testing.assert_array_equal(f // 2 + 1, a // 2 + 1)
    f[0, 0] = (7, 8, 9)                                     # once down it is an ordinary array: writable in place
    assert tuple(np.asarray(f)[0, 0]) == (7, 8, 9)
    np.bitwise_and(f, 0x0F, out=f)
    assert isinstance(f, DeviceFrame) and int(np.asarray(f).max()) <= 15


def test_any_pickle_is_the_host_array():
    from multiprocessing.reduction import ForkingPickler
    f, a, img = _frame(seed=2)
    for back in (pickle.loads(pickle.dumps(f)), copy.deepcopy(f), pickle.loads(bytes(ForkingPickler.dumps(f)))):
        assert type(back) is np.ndarray and back.dtype == np.uint8
        np.testing.assert_array_equal(back, a)
    assert img.ended == 1


def test_a_frame_nobody_read_still_ends_its_transfer_before_its_memory_is_reused():
    f, _, img = _frame(seed=3)
    del f
    assert img.ended == 1
