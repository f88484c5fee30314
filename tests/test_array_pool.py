"""transflow_amd.device.ArrayPool: an array -- or ANY VIEW of its memory -- that a caller still holds is never handed out
again.  Page-locked arrays are views themselves (frombuffer -> reshape) and numpy points a view of a view at the array that
owns the memory, so the pool must watch that owner too (advisor, round 4).  CPU: the page-locked allocation is stood in for
by a ctypes buffer with the same frombuffer -> reshape construction."""
import ctypes as C

import numpy as np
import pytest

from transflow_amd import device as D


def _like_pinned_empty(shape, dtype):
    dtype = np.dtype(dtype)
    n = int(np.prod(shape))
    buf = (C.c_char * max(1, n * dtype.itemsize))()
    return np.frombuffer(buf, dtype=dtype, count=n).reshape(shape)


@pytest.mark.parametrize("pinned", [False, True])
def test_a_held_view_keeps_its_array_out_of_the_pool(pinned, monkeypatch):
    monkeypatch.setattr(D, "pinned_empty", _like_pinned_empty)
    pool = D.ArrayPool((4, 5, 2), np.float32, limit=3, pinned=pinned)
    a = pool.take()
    ida = id(a)
    del a
    b = pool.take()
    assert id(b) == ida                       # free again: reused
    b[:] = 7
    u = b[..., 0]                             # what a caller keeps: a view, not the array
    del b
    for i in range(6):                        # more frames than the pool holds
        c = pool.take()
        assert not np.shares_memory(c, u)
        c[:] = i
    assert (u == 7).all()
    w = u.reshape(-1)[::2]                    # a view of the view
    del u
    c = pool.take()
    assert not np.shares_memory(c, w)
    del w, c
    assert len({id(pool.take()) for _ in range(5)}) == 1 and len(pool._arrays) <= 3
