"""CPU: the implicit-GEMM index math shared with the HIP kernels (opental_amd/csrc/conv_index.h),
run through naive loops (tests/cpu_conv_index.cpp, g++), against torch conv1d/conv3d + autograd
and against the oracle's SAME-padding rule.  Covers strides, asymmetric SAME pads, channel-sliced
(concat) tensors and level-packed pyramids."""
import ctypes
import os
import subprocess

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import afsd_oracle as O

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)


@pytest.fixture(scope="module")
def lib(tmp_path_factory):
    out = str(tmp_path_factory.mktemp("cpuconv") / "libcpuconv.so")
    subprocess.check_call(["g++", "-O2", "-shared", "-fPIC", "-std=c++17",
                           "-I" + os.path.join(REPO, "opental_amd", "csrc"),
                           os.path.join(HERE, "cpu_conv_index.cpp"), "-o", out])
    return ctypes.CDLL(out)


def geom(B, Cin, Cout, inn, k, s, nlev=1, lev=None):
    from opental_amd.common.conv_geom import make_geom
    return make_geom(B, Cin, Cout, inn, k, s, lev)


def run_case(lib, B, Cin, Cout, inn, k, s, lev=None, cslice=False):
    from opental_amd.common.conv_geom import make_geom
    rs = np.random.RandomState(B * 131 + Cin * 7 + Cout + sum(inn) + sum(k))
    g, outn = make_geom(B, Cin, Cout, inn, k, s, lev)
    x = torch.from_numpy(rs.randn(B, Cin, *inn).astype(np.float32))
    w = torch.from_numpy(rs.randn(Cout, Cin, *k).astype(np.float32))
    # torch reference with explicit SAME pad (per level when packed)
    def ref_fwd(xx, ww):
        if lev is None:
            pads = []
            for d in (2, 1, 0):
                f, b_ = O.same_pad(inn[d], k[d], s[d])
                pads += [f, b_]
            return F.conv3d(F.pad(xx, pads), ww, None, stride=s)
        outs = []
        for i in range(len(lev) - 1):
            seg = xx[:, :, lev[i]:lev[i + 1]]
            f, b_ = O.same_pad(seg.shape[2], k[0], 1)
            outs.append(F.conv3d(F.pad(seg, [0, 0, 0, 0, f, b_]), ww))
        return torch.cat(outs, 2)
    xr = x.clone().requires_grad_(True)
    wr = w.clone().requires_grad_(True)
    yr = ref_fwd(xr, wr)
    assert tuple(yr.shape[2:]) == tuple(outn), (yr.shape, outn)
    dy = torch.from_numpy(rs.randn(*yr.shape).astype(np.float32))
    yr.backward(dy)
    # strides: optionally place x / y inside wider (concat) buffers
    Pin, Pout = int(np.prod(inn)), int(np.prod(outn))
    xc, yc = (Cin + 3, Cout + 5) if cslice else (Cin, Cout)
    xbuf = torch.zeros(B, xc, *inn); ybuf = torch.full((B, yc, *outn), 7.0); dybuf = torch.zeros(B, yc, *outn)
    xo, yo = (2, 4) if cslice else (0, 0)
    xbuf[:, xo:xo + Cin] = x
    dybuf[:, yo:yo + Cout] = dy
    strides = (ctypes.c_int64 * 4)(xc * Pin, Pin, yc * Pout, Pout)
    garr = (ctypes.c_int * len(g))(*g)
    fp = lambda t, off=0: ctypes.c_void_p(t.data_ptr() + 4 * off)
    lib.cpu_conv_fwd(garr, strides, fp(xbuf, xo * Pin), fp(w), fp(ybuf, yo * Pout))
    assert torch.allclose(ybuf[:, yo:yo + Cout], yr.detach(), atol=2e-4, rtol=1e-4)
    if cslice:
        assert bool((ybuf[:, :yo] == 7).all()) and bool((ybuf[:, yo + Cout:] == 7).all())
    dxbuf = torch.full((B, xc, *inn), 7.0)
    lib.cpu_conv_dgrad(garr, strides, fp(dybuf, yo * Pout), fp(w), fp(dxbuf, xo * Pin))
    assert torch.allclose(dxbuf[:, xo:xo + Cin], xr.grad, atol=2e-4, rtol=1e-4)
    dw = torch.zeros_like(w)
    lib.cpu_conv_wgrad(garr, strides, fp(xbuf, xo * Pin), fp(dybuf, yo * Pout), fp(dw))
    assert torch.allclose(dw, wr.grad, atol=5e-4, rtol=1e-4)


CASES = [
    (2, 3, 4, (9, 8, 8), (7, 7, 7), (2, 2, 2)),      # Conv3d_1a style, odd T -> pad (3,3) / even -> (2,3)
    (1, 3, 5, (8, 10, 10), (7, 7, 7), (2, 2, 2)),
    (2, 4, 6, (5, 6, 6), (3, 3, 3), (1, 1, 1)),
    (1, 6, 3, (4, 3, 3), (1, 1, 1), (1, 1, 1)),
    (2, 5, 4, (6, 6, 6), (1, 6, 6), (1, 1, 1)),      # 'spatial_valid' projection: temporal SAME, spatial collapse
    (2, 5, 7, (16, 1, 1), (3, 1, 1), (1, 1, 1)),     # Unit1D k3 s1
    (2, 5, 7, (16, 1, 1), (3, 1, 1), (2, 1, 1)),     # Unit1D k3 s2 on even t -> right-only pad (0,1)
    (1, 4, 4, (7, 1, 1), (3, 1, 1), (2, 1, 1)),      # odd t, stride 2 -> pad (1,1)
    (3, 8, 2, (2, 1, 1), (3, 1, 1), (1, 1, 1)),      # t = 2 level
    (2, 4, 3, (5, 1, 1), (1, 1, 1), (1, 1, 1)),
]


@pytest.mark.parametrize("case", CASES)
def test_geometry_matches_torch(lib, case):
    run_case(lib, *case)


def test_channel_sliced_views(lib):
    run_case(lib, 2, 4, 6, (5, 6, 6), (3, 3, 3), (1, 1, 1), cslice=True)
    run_case(lib, 2, 5, 7, (16, 1, 1), (3, 1, 1), (2, 1, 1), cslice=True)


def test_spatial_valid_projection_has_no_spatial_pad(lib):
    from opental_amd.common.conv_geom import make_geom
    g, outn = make_geom(1, 4, 4, (8, 6, 6), (1, 6, 6), (1, 1, 1), spatial_valid=True)
    assert tuple(outn) == (8, 1, 1) and g[15:18] == [0, 0, 0]


def test_level_packed_conv1d(lib):
    lev = [0, 8, 12, 14, 15]
    run_case(lib, 2, 5, 6, (15, 1, 1), (3, 1, 1), (1, 1, 1), lev=lev)
    run_case(lib, 1, 3, 4, (15, 1, 1), (1, 1, 1), (1, 1, 1), lev=lev)


def test_same_pad_rule_matches_oracle(lib):
    f, o = ctypes.c_int(), ctypes.c_int()
    for size in range(1, 40):
        for k in (1, 2, 3, 6, 7):
            for s in (1, 2):
                lib.cpu_same_pad(size, k, s, ctypes.byref(f), ctypes.byref(o))
                fo, bo = O.same_pad(size, k, s)
                assert f.value == fo and o.value == (size + fo + bo - k) // s + 1


def test_fast_division_is_exact(lib):
    assert lib.cpu_fastdiv_check() == 0
    from opental_amd.common.conv_geom import make_geom
    for case in CASES:
        g, _ = make_geom(*case)
        garr = (ctypes.c_int * len(g))(*g)
        st = (ctypes.c_int64 * 4)(1, 1, 1, 1)
        assert lib.cpu_dec_fd_check(garr, st) == 0, case
