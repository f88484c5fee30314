"""CPU-side checks of the drop-in boundary: libtfhip.so loads without a GPU and
exports exactly the symbols include/tfhip.h declares; no compute is attempted."""
import ctypes
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    text = open(os.path.join(ROOT, "include", "tfhip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(tf_[a-z0-9_]+)\s*\(", text)))


@pytest.fixture(scope="module")
def lib():
    from transflow_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()
    return _lib


def test_header_and_library_agree(lib):
    names = header_symbols()
    assert len(names) >= 40
    assert sorted(lib.PROTOTYPES) == names
    dll = lib.load()
    for n in names:
        assert hasattr(dll, n), n
    out = subprocess.check_output(["nm", "-D", "--defined-only", lib.LIB_PATH], text=True)
    exported = sorted(l.split()[-1] for l in out.splitlines() if " T " in l)
    assert exported == names, "library exports symbols the header does not declare (or misses some)"


def test_load_makes_no_gpu_call_and_errors_are_reported(lib):
    dll = lib.load()
    assert dll.tf_abi_version() == 1
    n = ctypes.c_int(-1)
    rc = dll.tf_device_count(ctypes.byref(n))
    if rc != 0:  # no GPU here: the failure is reported, not swallowed
        assert rc == lib.TF_ERR_HIP and dll.tf_last_error()
        with pytest.raises(lib.TfError):
            lib.check(rc)


def test_status_to_exception_mapping(lib):
    dll = lib.load()
    assert dll.tf_device_count(None) == lib.TF_ERR_ARG
    with pytest.raises(ValueError):
        lib.check(lib.TF_ERR_ARG)
    with pytest.raises(IndexError):
        lib.check(lib.TF_ERR_INDEX)
    with pytest.raises(NotImplementedError):
        lib.check(lib.TF_ERR_UNSUPPORTED)


def test_code_object_targets_gfx950(lib):
    data = open(lib.LIB_PATH, "rb").read()
    assert b"gfx950" in data


def test_product_does_not_import_the_oracle():
    """The oracle is the checker: nothing the product ships may import, link or load it."""
    pkg = os.path.join(ROOT, "transflow_amd")
    pat = re.compile(r"(from|import)\s+oracle\b|oracle/|libfbref|remap_ref|farneback_ref")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")) or f == "Makefile":
                text = open(os.path.join(dirpath, f)).read()
                assert not pat.search(text), f"{f} references the oracle"
