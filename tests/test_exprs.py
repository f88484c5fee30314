"""The polar filter's expression compiler (transflow_amd/exprs.py) without a GPU: its programs are run
by a numpy stack machine that mirrors k_pp_polar step for step, and compared with what the reference
computed (tests/golden/flow_polar.npz).  Also pins the oracle's polar() on the same vectors."""
import os

import numpy as np
import pytest

from oracle import flow_ops_ref as F
from tests.helpers import GOLDEN
from transflow_amd.exprs import OPS, PolarFilter, Program, Unsupported

Z = np.load(os.path.join(GOLDEN, "flow_polar.npz"))
T = float(Z["t"])


def run_program(steps, r, a):
    """numpy twin of polar_eval in flowops.hip: a stack of float64 arrays, float32 steps round-trip."""
    st = []
    f32 = lambda x: np.asarray(x, np.float64).astype(np.float32)   # noqa: E731
    un = {"neg": np.negative, "sin": np.sin, "cos": np.cos, "tan": np.tan, "arcsin": np.arcsin, "arccos": np.arccos,
          "arctan": np.arctan, "sqrt": np.sqrt, "abs": np.abs, "exp": np.exp, "log": np.log, "log2": np.log2,
          "log10": np.log10, "floor": np.floor, "ceil": np.ceil, "rint": np.rint, "sign": np.sign,
          "square": np.square, "reciprocal": lambda x: 1 / x}
    bi = {"add": np.add, "sub": np.subtract, "mul": np.multiply, "div": np.divide, "pow": np.power, "mod": np.mod,
          "floordiv": np.floor_divide, "arctan2": np.arctan2, "minimum": np.minimum, "maximum": np.maximum,
          "hypot": np.hypot, "lt": np.less, "le": np.less_equal, "gt": np.greater, "ge": np.greater_equal,
          "eq": np.equal, "ne": np.not_equal}
    with np.errstate(all="ignore"):
        for op, wide, imm in steps:
            name = OPS[op]
            cast = (lambda x: np.asarray(x, np.float64)) if wide else f32
            if name == "push_r":
                st.append(np.asarray(r, np.float64))
            elif name == "push_a":
                st.append(np.asarray(a, np.float64))
            elif name == "push_const":
                st.append(np.float64(imm))
            elif name == "where":
                y, x, c = st.pop(), st.pop(), st.pop()
                st.append(np.where(c != 0, cast(x), cast(y)).astype(np.float64))
            elif name == "clip":
                hi, lo, x = st.pop(), st.pop(), st.pop()
                st.append(np.minimum(np.maximum(cast(x), cast(lo)), cast(hi)).astype(np.float64))
            elif name == "not":
                st.append((st.pop() == 0).astype(np.float64))
            elif name in bi:
                b, x = st.pop(), st.pop()
                st.append(np.asarray(bi[name](cast(x), cast(b)), np.float64))
            else:
                st.append(np.asarray(un[name](cast(st.pop())), np.float64))
    assert len(st) == 1
    return st[0]


def apply_polar(flow, er, ea, t):
    pf = PolarFilter(er, ea)
    sr, sa, wide_trig, wide_product = pf.programs(t)
    x, y = flow[:, :, 0], flow[:, :, 1]
    r = np.sqrt(x * x + y * y)
    a = np.arctan2(y, x)
    R, A = run_program(sr, r, a), run_program(sa, r, a)
    s, c = (np.sin(A), np.cos(A)) if wide_trig else (np.sin(A.astype(np.float32)), np.cos(A.astype(np.float32)))
    if wide_product:
        oy, ox = (R * s).astype(np.float32), (R * c).astype(np.float32)
    else:
        oy, ox = R.astype(np.float32) * s.astype(np.float32), R.astype(np.float32) * c.astype(np.float32)
    return np.stack(np.broadcast_arrays(ox, oy), axis=-1).astype(np.float32) + np.zeros_like(flow)


@pytest.mark.parametrize("i", range(int(Z["cases"])))
def test_compiled_programs_reproduce_the_reference(i):
    er, ea = str(Z[f"er_{i}"]), str(Z[f"ea_{i}"])
    out = apply_polar(Z[f"in_{i}"].copy(), er, ea, T)
    exp = Z[f"out_{i}"]
    # same numpy functions on the same float32 values, in the type numpy would have used: exact
    np.testing.assert_array_equal(out, exp, err_msg=f"{er!r} : {ea!r}")


@pytest.mark.parametrize("i", range(int(Z["cases"])))
def test_oracle_polar_golden(i):
    out = F.polar(Z[f"in_{i}"].copy(), str(Z[f"er_{i}"]), str(Z[f"ea_{i}"]), T)
    np.testing.assert_array_equal(out, Z[f"out_{i}"])


def test_compiler_limits_and_errors():
    assert Program("2*t+1").scalar_only and Program("2*t+1").host_value(0.5) == 2.0
    steps, kind = Program("r + numpy.float64(1)").resolve(0.0)
    assert kind == 1 and steps[-1][1] == 1                      # a numpy.float64 scalar is strong: float64 add
    steps, kind = Program("r + 1.5").resolve(0.0)
    assert kind == 0 and steps[-1][1] == 0                      # a Python float is weak: float32 add
    assert [OPS[s[0]] for s in Program("r**2").resolve(0.0)[0]] == ["push_r", "square"]
    assert [OPS[s[0]] for s in Program("r**0.5").resolve(0.0)[0]] == ["push_r", "sqrt"]
    assert [OPS[s[0]] for s in Program("r**t").resolve(3.0)[0]] == ["push_r", "push_const", "pow"]
    for bad in ("r.sum()", "numpy.cumsum(r)", "[r, a][0]", "r if t else a", "numpy.fft.fft(r)"):
        with pytest.raises(Unsupported):
            Program(bad)
    with pytest.raises(Unsupported):
        Program("+".join(["r"] * 40))                           # longer than the device's program
    with pytest.raises(Unsupported):
        Program("r + numpy.ones(3)").resolve(0.0)               # host subtree that is an array
    with pytest.raises(SyntaxError):
        Program("r +")
