"""oracle/flow_ops_ref.py against vectors the reference produced (tests/golden/flow_ops.npz)."""
import os

import numpy as np
import pytest

from oracle import flow_ops_ref as F
from tests.helpers import GOLDEN

Z = np.load(os.path.join(GOLDEN, "flow_ops.npz"))


def test_merge_golden():
    for i in range(int(Z["merge_cases"])):
        kind, n = str(Z[f"merge_{i}_kind"]), int(Z[f"merge_{i}_n"])
        out = F.merge(kind, [Z[f"merge_{i}_in{j}"] for j in range(n)])
        exp = Z[f"merge_{i}_out"]
        assert out.dtype == exp.dtype == np.float32, (kind, n)
        np.testing.assert_array_equal(out, exp, err_msg=f"{kind} n={n}")
    with pytest.raises(ValueError):
        F.merge("absmax", [Z["merge_0_in0"]] * 3)


def test_upscale_golden():
    for i in range(int(Z["up_cases"])):
        wf, hf = (int(v) for v in Z[f"up_{i}_f"])
        out = F.upscale(Z[f"up_{i}_in"], wf, hf)
        assert out.dtype == np.float32
        np.testing.assert_array_equal(out, Z[f"up_{i}_out"])


def test_kernel_post_process_golden():
    for i in range(int(Z["conv_cases"])):
        k, exp = Z[f"conv_{i}_kernel"], Z[f"conv_{i}_out"]
        out = F.post_process_with_kernel(Z[f"conv_{i}_in"], k, int(Z[f"conv_{i}_dir"]))
        assert out.dtype == exp.dtype, (i, k.dtype)
        np.testing.assert_array_equal(out, exp, err_msg=f"case {i} kernel {k.dtype}{k.shape}")


def test_render_golden():
    for i in range(int(Z["r1_cases"])):
        out = F.render1d(Z[f"r1_{i}_in"], float(Z[f"r1_{i}_scale"]), tuple(str(c) for c in Z[f"r1_{i}_colors"]),
                         bool(Z[f"r1_{i}_binary"]))
        np.testing.assert_array_equal(out, Z[f"r1_{i}_out"], err_msg=f"render1d {i}")
    for i in range(int(Z["r2_cases"])):
        out = F.render2d(Z[f"r2_{i}_in"], float(Z[f"r2_{i}_scale"]), tuple(str(c) for c in Z[f"r2_{i}_colors"]))
        np.testing.assert_array_equal(out, Z[f"r2_{i}_out"], err_msg=f"render2d {i}")
