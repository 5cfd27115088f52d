"""ctypes binding of libopental_hip.so (the C ABI declared in include/opental_hip.h).

The product path has NO CPU fallback: if the HIP library is missing or a call fails, a
RuntimeError is raised.  torch is used only for device memory and the current HIP stream.
"""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("OTAL_LIB_PATH") or os.path.join(_HERE, "lib", "libopental_hip.so")   # override: A/B kernel builds
ABI_VERSION = 24
F32, BF16, F16, F64 = 0, 1, 2, 3
E_UNSUPPORTED = -7      # OTAL_E_UNSUPPORTED: no kernel for this call's geometry (pair launches: issue the two launches instead)

_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} not found: build it with `python -m opental_amd.csrc.build` "
                "(hipcc --offload-arch=gfx950). There is no CPU fallback.")
        L = ctypes.CDLL(LIB_PATH)
        L.otal_error_string.restype = ctypes.c_char_p
        L.otal_error_string.argtypes = [ctypes.c_int]
        if L.otal_abi_version() != ABI_VERSION:
            raise RuntimeError("libopental_hip.so ABI version mismatch; rebuild")
        _lib = L
    return _lib


def set_option(name, value):
    """Run-time kernel-selection switch (include/opental_hip.h: otal_set_option); the C side reads the environment
    variable of the same name only once, at the switch's first use."""
    check(lib().otal_set_option(name.encode(), int(value)), "otal_set_option")


def check(rc, what):
    if rc != 0:
        raise RuntimeError(f"{what} failed: [{rc}] {lib().otal_error_string(rc).decode()}")


def ptr(t):
    return ctypes.c_void_p(t.data_ptr())


STREAM_OVERRIDE = None      # a raw stream handle: launches go there instead of torch's current stream (ops.SideWgrads)


def stream():
    """The raw HIP stream torch is currently launching on (side streams and graph capture included).  Goes through the
    C bindings directly: torch.cuda.current_stream() builds a Stream object (~10 us) and this runs ~110 times per step."""
    if STREAM_OVERRIDE is not None:
        return ctypes.c_void_p(STREAM_OVERRIDE)
    try:
        return ctypes.c_void_p(torch._C._cuda_getCurrentRawStream(torch._C._cuda_getDevice()))
    except AttributeError:          # private bindings moved: the public (slower) route
        return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def dtype_code(t):
    if t.dtype == torch.float32:
        return F32
    if t.dtype == torch.bfloat16:
        return BF16
    if t.dtype == torch.float16:
        return F16
    if t.dtype == torch.float64:
        return F64
    raise RuntimeError(f"unsupported dtype {t.dtype} (float32 / bfloat16 / float16 / float64)")


def require_device(*tensors):
    for t in tensors:
        if not t.is_cuda:
            raise RuntimeError("opental_amd ops run on the GPU only (tensor is on %s)" % t.device)
        if not t.is_contiguous():
            raise RuntimeError("tensor must be contiguous")


def int_array(values):
    return (ctypes.c_int * len(values))(*values)
