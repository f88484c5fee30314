"""The oracle's sum / static / introduction layers against vectors the reference produced
(tests/golden/layer2_*.npz, tools/capture_golden.py --layers2-only).  CPU only."""
import os

import numpy as np
import pytest

from oracle import remap_ref
from tests.helpers import capture_frame_numbers, layer2_case_files, oracle_layer2


@pytest.mark.parametrize("path", layer2_case_files(), ids=lambda p: os.path.basename(p)[7:-4])
def test_layer2_sequences_golden(path):
    z = np.load(path)
    layer = oracle_layer2(z)
    ns, cls = int(z["nsources"]), str(z["classname"])
    if "data_init" in z.files:
        np.testing.assert_array_equal(layer.data, z["data_init"])
    np.testing.assert_array_equal(np.asarray(layer.rgba), z["rgba_init"])
    for t in range(int(z["nframes"])):
        pixmaps = [z[f"pixmap_{s}"][t] for s in range(ns)]
        if cls == "introduction":
            # FakeSource.frame_number of the capture = number of next() calls so far - 1
            layer.update(z[f"flow_{t}"], pixmaps, frame_numbers=capture_frame_numbers(layer.prm, t, ns))
        else:
            layer.update(z[f"flow_{t}"], pixmaps, u=z[f"u_{t}"])
        if hasattr(layer, "data"):
            np.testing.assert_array_equal(layer.data, z[f"data_{t}"], err_msg=f"data, frame {t}")
        np.testing.assert_array_equal(np.asarray(layer.rgba), z[f"rgba_{t}"], err_msg=f"rgba, frame {t}")
        img = layer.render()
        frame = remap_ref.composite(np.broadcast_to(z["background"], (int(z["h"]), int(z["w"]), 3)), [img])
        np.testing.assert_array_equal(frame, z[f"frame_{t}"], err_msg=f"frame {t}")
        np.testing.assert_array_equal(np.asarray(layer.rgba), z[f"rgba_after_render_{t}"])
        if hasattr(layer, "data"):
            np.testing.assert_array_equal(layer.data, z[f"data_after_render_{t}"])


