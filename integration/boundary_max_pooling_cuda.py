"""boundary_max_pooling_cuda.py -- ctypes stand-in for the reference's CUDA extension.

Put this file on PYTHONPATH (next to, or instead of, the pybind module built from AFSD/prop_pooling/): the reference's
AFSD/prop_pooling/boundary_pooling_op.py:4 does `import boundary_max_pooling_cuda` and calls `forward(input, segments)`
/ `backward(grad_output, input, segments)` (boundary_max_pooling_cuda.cpp:52-55) -- nothing else in the reference
changes.  The library is found through OPENTAL_HIP_LIB or next to this repository's package.

Same contract as the extension: CUDA (HIP) tensors, contiguous (`TORCH_CHECK`s of .cpp:4-6 -> RuntimeError here), outputs
allocated by the wrapper, launch on the current stream, no host synchronisation.  float / double / half as the
reference dispatches (kernel.cu:99,:131) plus bfloat16.  `OTAL_BMP_COMPAT=1` reproduces the reference launcher's
tscale = N addressing in the backward (kernel.cu:121) instead of the mathematically correct gradient.
"""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
_PATH = os.environ.get("OPENTAL_HIP_LIB", os.path.join(_HERE, "..", "opental_amd", "lib", "libopental_hip.so"))
_lib = ctypes.CDLL(_PATH)
_lib.otal_error_string.restype = ctypes.c_char_p
_DT = {torch.float32: 0, torch.bfloat16: 1, torch.float16: 2, torch.float64: 3}


def _chk(rc, what):
    if rc != 0:
        raise RuntimeError(f"{what}: {_lib.otal_error_string(rc).decode()}")


def _p(t):
    return ctypes.c_void_p(t.data_ptr())


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _check(x, name):                                # CHECK_INPUT of boundary_max_pooling_cuda.cpp:4-6
    if not x.is_cuda:
        raise RuntimeError(f"{name} must be a CUDA tensor")
    if not x.is_contiguous():
        raise RuntimeError(f"{name} must be contiguous")


def forward(input, segments):                       # boundary_max_pooling_cuda.cpp:21-34
    _check(input, "input"); _check(segments, "segments")
    B, C, T = input.shape
    N = segments.shape[1]
    seg = segments if segments.dtype == torch.float32 else segments.float()     # (the kernel truncates them to int)
    out = torch.empty((B, C, N), dtype=input.dtype, device=input.device)
    _chk(_lib.otal_bmp_fwd(_p(input), _p(seg), _p(out), B, C, T, N, seg.shape[0], _DT[input.dtype], _stream()),
         "otal_bmp_fwd")
    return out


def backward(grad_output, input, segments):         # boundary_max_pooling_cuda.cpp:36-50
    _check(grad_output, "grad_output"); _check(input, "input"); _check(segments, "segments")
    B, C, T = input.shape
    N = segments.shape[1]
    seg = segments if segments.dtype == torch.float32 else segments.float()
    grad_input = torch.empty_like(input)            # every element is written by the kernel (no zero-fill launch needed)
    compat = 1 if os.environ.get("OTAL_BMP_COMPAT") == "1" else 0
    _chk(_lib.otal_bmp_bwd(_p(grad_output), _p(input), _p(seg), _p(grad_input), B, C, T, N, seg.shape[0], compat,
                           _DT[input.dtype], _stream()), "otal_bmp_bwd")
    return grad_input
