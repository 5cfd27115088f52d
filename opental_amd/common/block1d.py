"""ctypes side of the fused 1-D block launches (csrc/block1d.hip, `otal_b1d_*` in include/opental_hip.h).

A *problem* is one Unit1D + GroupNorm + ReLU block (AFSD/thumos14/BDNet.py:67-103,:129-203,:274-284) going forward, or the
GroupNorm / ReLU backward of a block fed by the data gradients of its consumers; a launch carries up to MAX_PROB problems.
Nothing here computes: tensors are torch device memory, the arithmetic runs in libopental_hip.so (no CPU fallback).
"""
import ctypes

import torch

from .. import _lib as L

MAX_SEG, MAX_ADD, MAX_PROB, MAX_RANGE, MAX_LEVELS = 4, 3, 6, 4, 8
FWD, BWD, PLAIN = 0, 1, 2


class Seg(ctypes.Structure):
    _fields_ = [("src", ctypes.c_void_p), ("wp", ctypes.c_void_p), ("src_bs", ctypes.c_int64), ("src_cs", ctypes.c_int64),
                ("src_elems", ctypes.c_int64), ("wp_elems", ctypes.c_int64), ("wp_pitch", ctypes.c_int),
                ("C", ctypes.c_int), ("kt", ctypes.c_int), ("mul", ctypes.c_int), ("off", ctypes.c_int),
                ("sgn", ctypes.c_int), ("shr", ctypes.c_int), ("par", ctypes.c_int), ("Tv", ctypes.c_int),
                ("use_levels", ctypes.c_int)]


class Add(ctypes.Structure):
    _fields_ = [("p", ctypes.c_void_p), ("bs", ctypes.c_int64), ("cs", ctypes.c_int64), ("Ta", ctypes.c_int),
                ("pad_", ctypes.c_int)]


class Problem(ctypes.Structure):
    _fields_ = [("seg", Seg * MAX_SEG), ("add", Add * MAX_ADD), ("nseg", ctypes.c_int), ("nadd", ctypes.c_int),
                ("B", ctypes.c_int), ("M", ctypes.c_int), ("cpg", ctypes.c_int), ("T", ctypes.c_int),
                ("kc", ctypes.c_int), ("epilogue", ctypes.c_int), ("nlev", ctypes.c_int),
                ("lev", ctypes.c_int * (MAX_LEVELS + 1)), ("nrange", ctypes.c_int),
                ("range_lev", ctypes.c_int * (MAX_RANGE + 1)), ("eps", ctypes.c_float), ("relu", ctypes.c_int),
                ("bias", ctypes.c_void_p), ("gamma", ctypes.c_void_p), ("beta", ctypes.c_void_p),
                ("c", ctypes.c_void_p), ("c_bs", ctypes.c_int64), ("c_cs", ctypes.c_int64),
                ("y", ctypes.c_void_p), ("y_bs", ctypes.c_int64), ("y_cs", ctypes.c_int64),
                ("stats", ctypes.c_void_p), ("partial", ctypes.c_void_p)]


_checked = False


def _check_layout():
    global _checked
    if not _checked:
        lib = L.lib()
        lib.otal_b1d_problem_bytes.restype = ctypes.c_size_t
        if lib.otal_b1d_problem_bytes() != ctypes.sizeof(Problem):
            raise RuntimeError("otal_b1d_problem layout mismatch between include/opental_hip.h and common/block1d.py")
        _checked = True


def _elems_from(t):
    """Elements readable from t's first element to the end of its storage (the kernel's buffer bound)."""
    return (t.untyped_storage().nbytes() - t.storage_offset() * t.element_size()) // t.element_size()


def _bs_cs(t):
    """(batch stride, channel stride) of a (B,C,T) tensor whose positions are dense."""
    B, C, T = t.shape
    if T > 1 and t.stride(2) != 1:
        raise RuntimeError("block1d: positions must be dense")
    cs = t.stride(1) if C > 1 else T
    bs = t.stride(0) if B > 1 else cs * C
    return bs, cs


class Pack:
    """bf16 operand packs of one weight (Cout,Cin,kt): `fwd` rows = Cout, `dgrad` rows = Cin (csrc/block1d.hip)."""

    def __init__(self, w, fwd=True, dgrad=True):
        self.w = w
        co, ci, kt = w.shape
        if co % 8 or ci % 8:
            raise RuntimeError("block1d pack: channel counts must be multiples of 8")
        self.fwd = torch.empty((co, ci * kt), dtype=torch.int16, device=w.device) if fwd else None
        self.dgrad = torch.empty((ci, co * kt), dtype=torch.int16, device=w.device) if dgrad else None


class PackSet:
    """All packs of a model; refresh() re-packs them from the current weights in ONE launch (the item table lives on the
    device: build it outside a graph capture, i.e. by calling refresh() once eagerly)."""

    def __init__(self, packs):
        self.packs = list(packs)
        self._table = None
        self._key = None

    def _build(self):
        recs, first = [], 0
        for p in self.packs:
            co, ci, kt = p.w.shape
            recs.append((p.w.data_ptr(), p.fwd.data_ptr() if p.fwd is not None else 0,
                         p.dgrad.data_ptr() if p.dgrad is not None else 0, co, ci, kt, first))
            first += (co * ci * kt // 8 + 255) // 256
        import struct
        raw = b"".join(struct.pack("<QQQiiii", *r) for r in recs)
        host = torch.frombuffer(bytearray(raw), dtype=torch.uint8)
        self._table = host.to(self.packs[0].w.device)
        self._blocks = first

    def refresh(self):
        key = tuple(p.w.data_ptr() for p in self.packs)
        if key != self._key:
            if torch.cuda.is_current_stream_capturing():
                raise RuntimeError("block1d: the pack table must be built before the step is captured")
            self._build()
            self._key = key
        L.check(L.lib().otal_b1d_pack(L.ptr(self._table), len(self.packs), int(self._blocks), L.stream()), "otal_b1d_pack")


def seg(src, wp, C, kt, mul=1, off=0, sgn=1, shr=0, par=0, Tv=0, use_levels=False, row0=0):
    """One K segment: `src` (B,C,Ts) fp32, `wp` a Pack tensor (rows, pitch); row0 = first pack row of the problem."""
    bs, cs = _bs_cs(src)
    s = Seg()
    s.src = src.data_ptr()
    wrow = wp[row0:]
    s.wp = wrow.data_ptr()
    s.src_bs, s.src_cs = bs, cs
    s.src_elems = _elems_from(src)
    s.wp_elems = _elems_from(wrow)
    s.wp_pitch = wp.stride(0)
    s.C, s.kt, s.mul, s.off, s.sgn, s.shr, s.par, s.Tv, s.use_levels = C, kt, mul, off, sgn, shr, par, Tv, int(use_levels)
    return s, (src, wp)


def problem(epilogue, B, M, T, segs, y, c=None, stats=None, gamma=None, beta=None, bias=None, adds=(), partial=None,
            levels=None, ranges=None, cpg=None, groups=32, kc=None, eps=1e-5, relu=True):
    """Fill an otal_b1d_problem.  levels: level table over [0,T) (None: one level); ranges: level indices where the
    workgroups of a (sample, group) split (None: one workgroup)."""
    P = Problem()
    keep = []
    for i, (s, k) in enumerate(segs):
        P.seg[i] = s
        keep.append(k)
    P.nseg = len(segs)
    for i, (t, Ta) in enumerate(adds):
        bs, cs = _bs_cs(t)
        P.add[i].p, P.add[i].bs, P.add[i].cs, P.add[i].Ta = t.data_ptr(), bs, cs, (T if Ta is None else Ta)
        keep.append(t)
    P.nadd = len(adds)
    P.B, P.M, P.T = B, M, T
    P.cpg = cpg if cpg is not None else M // groups
    lev = [0, T] if levels is None else list(levels)
    P.nlev = len(lev) - 1
    for i in range(MAX_LEVELS + 1):
        P.lev[i] = lev[min(i, P.nlev)]
    rg = [0, P.nlev] if ranges is None else list(ranges)
    P.nrange = len(rg) - 1
    for i in range(MAX_RANGE + 1):
        P.range_lev[i] = rg[min(i, P.nrange)]
    nmax = max(lev[rg[i + 1]] - lev[rg[i]] for i in range(P.nrange))
    P.kc = kc if kc is not None else (128 if nmax <= 128 else 64)
    P.epilogue, P.eps, P.relu = epilogue, eps, int(relu)
    opt = lambda t: t.data_ptr() if t is not None else None
    P.bias, P.gamma, P.beta = opt(bias), opt(gamma), opt(beta)
    if c is not None:
        P.c = c.data_ptr()
        P.c_bs, P.c_cs = _bs_cs(c)
    P.y = y.data_ptr()
    P.y_bs, P.y_cs = _bs_cs(y)
    P.stats, P.partial = opt(stats), opt(partial)
    keep += [y, c, stats, gamma, beta, bias, partial]
    return P, keep


def launch(problems):
    """problems: list of (Problem, keep) from problem(); returns the error code of OTAL_E_UNSUPPORTED unraised."""
    _check_layout()
    n = len(problems)
    arr = (Problem * n)(*[p for p, _ in problems])
    rc = L.lib().otal_b1d_launch(arr, n, L.stream())
    if rc == L.E_UNSUPPORTED:
        return False
    L.check(rc, "otal_b1d_launch")
    return True
