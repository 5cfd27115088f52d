"""CPU: the C-ABI library loads and exports every symbol include/opental_hip.h declares
(no compute calls -- there is no GPU here), and argument errors come back as negative codes."""
import ctypes
import os
import re

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    names = []
    for fn in sorted(os.listdir(os.path.join(REPO, "include"))):
        if fn.endswith(".h"):
            txt = open(os.path.join(REPO, "include", fn)).read()
            txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
            names += re.findall(r"\b(otal_[a-z0-9_]+)\s*\(", txt)
    return sorted(set(names))


@pytest.fixture(scope="module")
def lib():
    from opental_amd.csrc import build
    path = build.LIB
    if not os.path.exists(path):
        build.build(verbose=False)
    return ctypes.CDLL(path)


def test_every_declared_symbol_is_exported(lib):
    names = declared_symbols()
    assert len(names) >= 6
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, missing


def test_abi_version_and_error_strings(lib):
    assert lib.otal_abi_version() == 24
    lib.otal_error_string.restype = ctypes.c_char_p
    assert b"batch" in lib.otal_error_string(-4)
    assert lib.otal_error_string(0) == b"success"


def test_argument_errors_do_not_launch(lib):
    one = ctypes.c_void_p(16)  # never dereferenced: argument checks come first
    f = ctypes.cast(one, ctypes.POINTER(ctypes.c_float))
    assert lib.otal_bmp_fwd(None, f, one, 1, 4, 8, 2, 1, 0, None) == -1      # null
    assert lib.otal_bmp_fwd(one, f, one, 1, 5, 8, 2, 1, 0, None) == -3       # odd C
    assert lib.otal_bmp_fwd(one, f, one, 2, 4, 8, 2, 1, 0, None) == -4       # H3: batch mismatch
    assert lib.otal_bmp_fwd(one, f, one, 1, 4, 8, 2, 1, 7, None) == -5       # dtype
    assert lib.otal_bmp_fwd(one, f, one, 1, 4, 0, 2, 1, 0, None) == -2       # shape
    assert lib.otal_bmp_bwd(one, one, f, one, 1, 4, 8, 16, 1, 1, 0, None) == -2  # compat with N > T


def test_product_does_not_import_oracle():
    bad = []
    for root, _, files in os.walk(os.path.join(REPO, "opental_amd")):
        for fn in files:
            if fn.endswith(".py"):
                src = open(os.path.join(root, fn)).read()
                if re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M) or "from .. import oracle" in src:
                    bad.append(os.path.join(root, fn))
    assert not bad, bad
