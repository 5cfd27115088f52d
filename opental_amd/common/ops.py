"""Thin host wrappers over the C ABI (include/opental_hip.h).  torch is used for device memory
and the current HIP stream only; every numeric step below runs in libopental_hip.so.

Tensors may be channel slices of a larger contiguous buffer (``buf[:, c0:c1]``): batch and
channel strides are passed to the kernels, which is how Inception concatenation is written
in place instead of through torch.cat.
"""
import bisect
import ctypes
import os
import warnings

import torch

from .. import _lib as L
from .conv_geom import make_geom

_WORKSPACE = {}
WORKSPACE_BYTES = 192 << 20         # what one launch is offered (its split-K choice depends on it: kept fixed)
WORKSPACE_TOTAL = 512 << 20         # the buffer: room for the slabs of deferred weight-gradient reductions in front of that

# Optional per-launch timing of the implicit-GEMM kernels (bench.py's roofline leg): when a list is
# installed here, every conv launch is bracketed by HIP events recorded on the launch stream
# (torch's current stream IS the stream the C ABI launches on) and tagged with its algorithmic FLOPs.
CONV_PROFILE = None

# 0: fp32 MFMA (bit-for-bit an fp32 fma chain; the parity path).  1: bf16 MFMA operands with fp32
# accumulation (BASELINE.json's benchmark dtype; tensors stay fp32 in HBM).  Set by bench.py --dtype.
CONV_PRECISION = 0
# bf16 STORAGE of the largest activations in the bf16-operand mode (Conv3d_1a's output and its gradient: the tensors are
# only ever consumed through bf16 roundings, so the forward values do not change).  False: fp32 tensors everywhere.
HALF_STORAGE = os.environ.get("OTAL_HALF_STORAGE", "1") != "0"
# ... and of the outputs of direct 3x3x3 layers that only feed a strided pool (Conv3d_2c -> MaxPool3d_3a).  OFF by default:
# measured 11.66 vs 11.68 ms per step (within noise: the layer is MFMA/LDS-bound, not store-bound), while the pool ties the
# rounding creates move Conv3d_1a's weight gradient further (cosine 0.989 vs 0.997 with the fp32-stored run).
HALF_ACT_DIRECT = os.environ.get("OTAL_HALF_ACT_DIRECT", "0") != "0"
# ... and (round 4) of EVERY activation and data gradient between Conv3d_1a and the last backbone module the bf16-tensor kernels
# cover (Mixed_4f): precision bit 3 of the C ABI, otal_maxpool3d_*_io (common/i3d_backbone.py: "bf16 STORAGE").  The forward
# values do not change; gradients change only where a tensor has two producers (one more bf16 rounding).
HALF_CHAIN = os.environ.get("OTAL_HALF_CHAIN", "1") != "0"


def _prof_begin():
    if CONV_PROFILE is None:
        return None
    ev = torch.cuda.Event(enable_timing=True)
    ev.record()
    return ev


def _prof_end(ev, mode, g, problems=1):
    if ev is None:
        return
    end = torch.cuda.Event(enable_timing=True)
    end.record()
    B, Cin, Cout = g[0], g[1], g[2]
    flops = 2.0 * B * Cout * g[6] * g[7] * g[8] * Cin * g[9] * g[10] * g[11] * problems
    CONV_PROFILE.append((mode, flops, ev, end))


def workspace(device, side=False):
    """One split-K scratch buffer per device and stream (the main stream's, and the weight-gradient side stream's --
    SideWgrads); launches on a stream are serialised, so sharing within one is safe."""
    key = (device.type, device.index, side)
    ws = _WORKSPACE.get(key)
    if ws is None:
        ws = torch.empty(WORKSPACE_TOTAL, dtype=torch.uint8, device=device)
        _WORKSPACE[key] = ws
    return ws


# ---- deferred weight-gradient reductions (include/opental_hip.h: otal_conv_defer_reduces).  While a trainer's backward
# runs, the split-K reduce of a weight gradient is recorded instead of launched; its slabs stay in the workspace and every
# later launch on that stream works BEHIND them (_WS_CURSOR); flush_reduces() runs all recorded reductions as one launch --
# called when the backbone announces a module's gradients (grads_ready), at the trainer's bucket flushes, and when the
# slabs fill up.  The library keeps ONE list of recorded reductions, so all of them sit on one stream at a time
# (_DEFER_OWNER: (side workspace?, torch stream or None = the current one)); a weight gradient issued on the other stream
# flushes them first.
_DEFER = False
_WS_CURSOR = [0, 0, 0, 0]   # main / weight-gradient side / branch lane / early lane workspace
_WS_SIDE = 0                # 1 while SideWgrads issues a launch on its stream, 2 inside a BranchLane block
_SIDE_NOW = None            # ... and that torch stream
_DEFER_OWNER = None


def _ws_args(device):
    """(pointer, size) of the workspace a launch may use now."""
    ws = workspace(device, _WS_SIDE)
    cur = _WS_CURSOR[_WS_SIDE]
    return ctypes.c_void_p(ws.data_ptr() + cur), ctypes.c_size_t(min(ws.numel() - cur, WORKSPACE_BYTES))


def defer_reduces(on):
    global _DEFER
    if not on:
        flush_reduces()
    L.check(L.lib().otal_conv_defer_reduces(int(bool(on))), "otal_conv_defer_reduces")
    _DEFER = bool(on)


def _stream_wait(stream):
    """torch's current stream waits for what `stream` has been given so far (works inside a graph capture)."""
    L.check(L.lib().otal_stream_wait(L.stream(), ctypes.c_void_p(stream.cuda_stream)), "otal_stream_wait")


def flush_reduces(wait=True):
    """Run the recorded reductions on the stream their slabs were written on; when that is not the stream the caller
    launches on, the caller's stream waits for them (unless wait=False: the caller joins the side stream later)."""
    global _DEFER_OWNER
    own = _DEFER_OWNER
    if _DEFER and own is not None:
        stream = own[1]
        if stream is None:
            L.check(L.lib().otal_conv_flush_reduces(L.stream()), "otal_conv_flush_reduces")
        else:
            L.check(L.lib().otal_conv_flush_reduces(ctypes.c_void_p(stream.cuda_stream)), "otal_conv_flush_reduces")
            if wait and L.STREAM_OVERRIDE != stream.cuda_stream:
                _stream_wait(stream)
    _DEFER_OWNER = None
    _WS_CURSOR[0] = _WS_CURSOR[1] = 0           # (the branch lane never holds recorded reductions: its cursor stays 0)


def _after_wgrad(device):
    """Advance the workspace cursor past the slabs a deferred reduction still needs; flush when they pile up."""
    global _DEFER_OWNER
    lib = L.lib()
    lib.otal_conv_deferred_end.restype = ctypes.c_size_t
    end = int(lib.otal_conv_deferred_end())
    if end:
        _DEFER_OWNER = (_WS_SIDE, _SIDE_NOW)
        _WS_CURSOR[_WS_SIDE] = (end - workspace(device, _WS_SIDE).data_ptr() + 255) & ~255
        if _WS_CURSOR[_WS_SIDE] > WORKSPACE_TOTAL - WORKSPACE_BYTES:
            flush_reduces()


# ---- weight gradients on a second HIP stream.  The weight gradient of a layer and its data gradient only share their
# INPUT (dy): the data-gradient chain is the critical path of the backward pass, the weight gradients hang off it, and
# nothing on the main stream reads them before the trainer's bucket flush / optimizer step.  Most of the layers work on
# 6x6 / 3x3 planes or 1-D maps and cannot fill 256 CUs on their own (split-K, ~20-40 us launches at 50-250 TFLOP/s), so the
# two families run on two streams and share the chip.  Results are bit-identical: same kernels, same order within each
# family.
WGRAD_STREAM = os.environ.get("OTAL_WGRAD_STREAM", "1") != "0"
# True while a trainer's single-use backward runs (DetectorTrainer.begin_backward(early=True)): weight gradients written
# into their arena slots are joined by the trainer (bucket flushes of a data-parallel run, end_backward), not at the end
# of each autograd node.
SIDE_DEFER_JOIN = False
SIDE_IN_GRAPH = os.environ.get("OTAL_WGRAD_STREAM_IN_GRAPH", "0") != "0"      # experiments only
_SIDES = {}


class LanePlan:
    """A training step captured as a SEQUENCE of HIP graphs on two lanes.  A replayed hipGraph runs its branches one after
    the other, so a fork inside one captured graph buys nothing; graphs launched on two STREAMS do run side by side.  The
    step is therefore cut wherever SideWgrads issues a chunk of weight gradients: the main lane's launches up to the cut
    are one graph, the chunk is a second graph for the side stream, and the main lane continues in a third.  Replay: main
    graphs go to the caller's stream in order; a side graph is launched on the side stream behind the main graph that
    precedes it (one event) and runs beside the main graphs that follow; ("join",) makes the main stream wait for the side
    stream; ("call", fn) entries are host actions between graphs (collectives of a data-parallel run).  All graphs share
    one memory pool: the captures never overlap, and every tensor a side graph reads is kept alive until the capture is
    complete (`keep`), so no later main-lane allocation can land on it."""

    def __init__(self, side_stream):
        self.entries = []
        self.pool = None
        self.cur = None
        self.side = side_stream
        self.keep = []
        self.mark_event = None      # ("mark",) records it on the side stream, ("wait_mark",) makes the main stream wait for it

    def begin_main(self):
        g = torch.cuda.CUDAGraph()
        if self.pool is None:
            self.pool = torch.cuda.graph_pool_handle()
        g.capture_begin(pool=self.pool, capture_error_mode="relaxed")
        self.cur = g

    def end_main(self):
        if self.cur is not None:
            with warnings.catch_warnings():         # two cuts in a row leave an empty graph between them: harmless
                warnings.simplefilter("ignore")
                self.cur.capture_end()
            self.entries.append(("main", self.cur))
            self.cur = None

    def side_chunk(self, run):
        self.end_main()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.stream(self.side):
            g.capture_begin(pool=self.pool, capture_error_mode="relaxed")
            try:
                run()
            finally:
                g.capture_end()
        self.entries.append(("side", g))
        self.begin_main()

    def cut(self, entry):
        """A host action between two main graphs."""
        self.end_main()
        self.entries.append(entry)
        self.begin_main()

    def replay(self):
        lib, main = L.lib(), L.stream()
        side_raw = ctypes.c_void_p(self.side.cuda_stream)
        for e in self.entries:
            kind = e[0]
            if kind == "main":
                e[1].replay()
            elif kind == "side":
                L.check(lib.otal_stream_wait(side_raw, main), "otal_stream_wait")
                with torch.cuda.stream(self.side):
                    e[1].replay()
            elif kind == "join":
                L.check(lib.otal_stream_wait(main, side_raw), "otal_stream_wait")
            elif kind == "mark":            # the side lane's position NOW: every chunk issued so far, none of the later ones
                if self.mark_event is None:
                    self.mark_event = torch.cuda.Event()
                self.mark_event.record(self.side)
            elif kind == "wait_mark":
                torch.cuda.current_stream().wait_event(self.mark_event)
            else:
                e[1]()


LANES = None        # the LanePlan being captured (DetectorTrainer.capture_step(lanes=True)), or None


class SideWgrads:
    """wgrad(...) RECORDS conv_wgrad(...) and returns its destination; issue() launches what has been recorded on the side
    stream, behind everything the main stream has been given so far (one fork per chunk of layers: a fork is two HIP calls,
    and switching streams between consecutive launches costs the runtime more than staying on one), followed by ONE launch
    for the chunk's recorded split-K reductions; join() issues and makes the main (= torch's current) stream wait.  x and
    dy are kept alive until join(): the caching allocator would otherwise hand their blocks to later main-stream launches
    while the side stream still reads them.  Nothing on the main stream may modify a recorded dy before join() -- the
    backward passes below never write to a gradient tensor after its producer."""
    CHUNK = int(os.environ.get("OTAL_WGRAD_CHUNK", "4"))

    def __init__(self, device):
        prio = int(os.environ.get("OTAL_WGRAD_STREAM_PRIORITY", "0"))
        self.side = torch.cuda.Stream(device=device, priority=prio)
        self._raw = ctypes.c_void_p(self.side.cuda_stream)
        self.pending = []
        self.keep = []

    @property
    def on(self):
        # not inside a graph capture: a replayed hipGraph runs its branches one after the other (measured at b = 1, 2, 8:
        # 0.06-0.2 ms per step SLOWER with the fork than without), so only eager launches gain from the second stream
        return WGRAD_STREAM and CONV_PROFILE is None and (LANES is not None or SIDE_IN_GRAPH
                                                          or not torch.cuda.is_current_stream_capturing())

    def wgrad(self, x, dy, w_shape, k, s, spatial_valid=False, levels=None, out=None):
        if not self.on:
            return conv_wgrad(x, dy, w_shape, k, s, spatial_valid=spatial_valid, levels=levels, out=out)
        if out is None:
            out = torch.empty(tuple(w_shape), dtype=torch.float32, device=x.device)
        # the recorded call writes through an ALIAS: a second reference to the returned tensor itself would stop autograd
        # from adopting it as the parameter's .grad (AccumulateGrad clones a gradient somebody else still holds -- here a
        # clone of memory the weight gradient has not been written to yet)
        dst = out.detach()
        self.pending.append(lambda: conv_wgrad(x, dy, w_shape, k, s, spatial_valid=spatial_valid, levels=levels, out=dst))
        self.keep.append((x, dy))
        return out

    def wgrad_pair(self, xs, dys, w_shape, k, s, levels=None, outs=(None, None)):
        """Two sibling layers: the pair launch, or two launches where the library has no pair kernel."""
        if not self.on:
            r = conv_wgrad_pair(xs, dys, w_shape, k, s, levels, outs)
            return r if r is not None else [conv_wgrad(x, dc, w_shape, k, s, levels=levels, out=o) for x, dc, o in zip(xs, dys, outs)]
        outs = [o if o is not None else torch.empty(tuple(w_shape), dtype=xs[0].dtype, device=xs[0].device) for o in outs]
        dsts = [o.detach() for o in outs]       # aliases, as in wgrad()

        def run():
            if conv_wgrad_pair(xs, dys, w_shape, k, s, levels, dsts) is None:
                for x, dc, o in zip(xs, dys, dsts):
                    conv_wgrad(x, dc, w_shape, k, s, levels=levels, out=o)
        self.pending.append(run)
        self.keep.append((xs, dys))
        return outs

    def issue(self):
        if not self.pending:
            return
        if _DEFER_OWNER is not None and not _DEFER_OWNER[0]:
            flush_reduces()                 # reductions recorded on the main stream: run them there first
        raw = self._raw

        def run():
            global _WS_SIDE, _SIDE_NOW
            _WS_SIDE, _SIDE_NOW, L.STREAM_OVERRIDE = 1, self.side, raw.value
            try:
                for fn in self.pending:
                    fn()
                if _DEFER and _DEFER_OWNER is not None:
                    flush_reduces(wait=False)   # this chunk's reductions: one launch, on the side stream
            finally:
                _WS_SIDE, _SIDE_NOW, L.STREAM_OVERRIDE = 0, None, None
                self.pending.clear()
        if LANES is not None:               # capture: the chunk becomes a graph of its own (LanePlan)
            LANES.side_chunk(run)
            return
        L.check(L.lib().otal_stream_wait(raw, L.stream()), "otal_stream_wait")      # torch's current stream: the dy producers are on it
        run()

    def flush(self):
        """A good place to cut (a backbone module is complete): issue when a chunk's worth of layers is waiting."""
        if len(self.pending) >= self.CHUNK:
            self.issue()

    def join(self):
        if self.keep:
            self.issue()
            if LANES is not None:
                LANES.cut(("join",))
                LANES.keep.extend(self.keep)
            else:
                _stream_wait(self.side)
            self.keep.clear()

    def node_end(self, in_slots):
        """End of an autograd node: its weight gradients are handed to autograd, which may add them to earlier ones on the
        main stream -- unless they sit in the trainer's arena slots and the trainer joins (SIDE_DEFER_JOIN)."""
        if not (SIDE_DEFER_JOIN and in_slots):
            self.join()
        else:
            self.flush()


# ---- a third stream for independent branches of the main lane (the Inception modules' small branches run beside the large
# one: common/i3d_backbone.py).  with lane: ... issues this library's launches on the lane's stream, with the lane's own
# workspace; fork() / join() are one event each.
BRANCH_LANE = os.environ.get("OTAL_BRANCH_LANE", "1") != "0"
# the pyramid as two hand-scheduled autograd nodes (thumos14/pyramid_fused.py) instead of one node per block
FUSED_PYRAMID = os.environ.get("OTAL_FUSED_PYRAMID", "1") != "0"
PYRAMID_LANE = os.environ.get("OTAL_PYRAMID_LANE", "1") != "0"        # ... with their independent chains on the branch lane
_BRANCHES = {}


class BranchLane:
    def __init__(self, device, ws_index=2):
        self.stream = torch.cuda.Stream(device=device)
        self._raw = ctypes.c_void_p(self.stream.cuda_stream)
        self._saved = None
        self._ws = ws_index         # the lane's own split-K workspace (launches of two lanes run side by side)

    @property
    def on(self):
        return BRANCH_LANE and CONV_PROFILE is None

    def fork(self):
        """The lane continues behind everything torch's current stream has been given so far."""
        L.check(L.lib().otal_stream_wait(self._raw, L.stream()), "otal_stream_wait")

    def join(self):
        L.check(L.lib().otal_stream_wait(L.stream(), self._raw), "otal_stream_wait")

    def __enter__(self):
        global _WS_SIDE
        self._saved = (_WS_SIDE, L.STREAM_OVERRIDE)
        _WS_SIDE, L.STREAM_OVERRIDE = self._ws, self._raw.value
        # the lane is also torch's current stream inside the block: tensors allocated here come from the LANE's pool of the
        # caching allocator.  With the main stream current, a block the main lane had just freed (its last reader still
        # queued there) could be handed to a lane tensor and overwritten by a lane kernel first -- two lanes, one pool
        self._ctx = torch.cuda.stream(self.stream)
        self._ctx.__enter__()
        return self

    def __exit__(self, *exc):
        global _WS_SIDE
        self._ctx.__exit__(*exc)
        self._ctx = None
        _WS_SIDE, L.STREAM_OVERRIDE = self._saved
        return False


def branch_lane(device):
    key = (device.type, device.index)
    b = _BRANCHES.get(key)
    if b is None:
        b = _BRANCHES[key] = BranchLane(device)
    return b


# ---- the early lane: the Mixed_4f projection of the pyramid (Unit3D [1,6,6], AFSD/thumos14/BDNet.py:129-139,:310-313) needs
# only Mixed_4f, which the backbone finishes ~280 us before Mixed_5c -- five small-plane launches that leave most of the chip
# idle.  The backbone calls ENDPOINT_HOOKS[name](tensor) where an endpoint is final; the pyramid's hook runs the projection
# on this lane (its own stream and split-K workspace: the branch lane forks and joins inside Mixed_5b / 5c meanwhile) and
# parks the result in EARLY_RESULTS for TrunkFunction.forward (thumos14/pyramid_fused.py), which joins the lane.  Backward,
# mirrored: the projection's data gradient (the gradient of Mixed_4f) runs on the lane while the main lane already walks
# Mixed_5c / 5b; the backbone joins (join_pending) where it picks that gradient up.
EARLY_PROJ = os.environ.get("OTAL_EARLY_PROJ", "1") != "0"
ENDPOINT_HOOKS = None
EARLY_RESULTS = {}
PENDING_JOINS = {}
_EARLY = {}


def early_lane(device):
    key = (device.type, device.index)
    b = _EARLY.get(key)
    if b is None:
        b = _EARLY[key] = BranchLane(device, ws_index=3)
    return b


def join_pending(t):
    """The main lane waits for the lane that still writes `t` (a gradient handed over without a join)."""
    lane = PENDING_JOINS.pop(t.data_ptr(), None) if PENDING_JOINS else None
    if lane is not None:
        lane.join()


_CAPTURE_STREAMS = {}


def capture_stream(device):
    """ONE stream per device for every capture / warm-up of every trainer: a fresh torch.cuda.Stream() per capture is a
    fresh HIP stream that keeps a share of one of the runtime's four hardware queues for the life of the process -- enough
    of them and a lane of a later step lands on the queue of another (measured with tools/probe_fed.py: the same replayed
    step at 981 or 854 clips/s depending on how many streams existed before its lanes were created)."""
    key = (device.type, device.index)
    st = _CAPTURE_STREAMS.get(key)
    if st is None:
        st = _CAPTURE_STREAMS[key] = torch.cuda.Stream(device=device)
    return st


def side_wgrads(device):
    key = (device.type, device.index)
    sd = _SIDES.get(key)
    if sd is None:
        sd = _SIDES[key] = SideWgrads(device)
    return sd


# ---- the optimizer step beside the LAST weight gradients.  The stem's weight gradients (Conv3d_2c, 2b, 1a: ~0.75 ms on the
# side lane) are the step's tail: the main lane has nothing left but Adam, which needs every gradient -- except that an
# elementwise optimizer can update all OTHER parameters as soon as THEIR gradients are final.  The backbone's backward
# calls late_mark() where only its stem tail is left; in a lane-graph capture that closes the side lane's chunk and records a
# ("mark",) entry, and the trainer (DetectorTrainer.end_backward) runs Adam over everything but LATE_WEIGHTS behind the mark,
# beside the tail's weight gradients, and Adam over the tail's few parameters after the final join.
EARLY_ADAM = os.environ.get("OTAL_EARLY_ADAM", "1") != "0"
# The backbone's last weight gradient (Conv3d_1a, 0.44 ms) on the MAIN lane, which has nothing left when it reaches it, while
# the weight-gradient lane works off what it still holds (~0.5 ms: the end of Mixed_3, Conv3d_2c, 2b).  Measured +-0 in the
# first session of round 6 (8.217 vs 8.228 ms: the kernel then kept three workgroups per CU resident and the two lanes'
# kernels took turns); with the second session's kernel (two per CU, 220 registers) 8.07 -> 7.99 ms, 8.25 -> 8.13 on a slower box.
# Also moving Conv3d_2b's or 2c's behind it: no further gain (8.02-8.07).
LAST_WGRAD_MAIN = os.environ.get("OTAL_LAST_WGRAD_MAIN", "1") != "0"
LATE_WEIGHTS = None


def late_mark(weights):
    global LATE_WEIGHTS
    if LANES is None or not EARLY_ADAM or LATE_WEIGHTS is not None:
        return
    side_issue()
    LANES.cut(("mark",))
    LATE_WEIGHTS = list(weights)


def take_late_weights():
    global LATE_WEIGHTS
    w, LATE_WEIGHTS = LATE_WEIGHTS, None
    return w


def side_join():
    for sd in _SIDES.values():
        sd.join()


def side_issue():
    for sd in _SIDES.values():
        sd.issue()


def side_busy():
    """True when weight gradients have been given to a side stream since its last join."""
    return any(sd.keep for sd in _SIDES.values())


def _as5(t):
    if t.dim() == 3:
        return t.unsqueeze(-1).unsqueeze(-1)
    if t.dim() != 5:
        raise RuntimeError("expected a (B,C,T) or (B,C,T,H,W) tensor")
    return t


def _check(t5, name, dtype=torch.float32):
    if not t5.is_cuda:
        raise RuntimeError(f"{name}: opental_amd ops run on the GPU only (no CPU fallback)")
    if t5.dtype != dtype:
        raise RuntimeError(f"{name}: {dtype} expected, got {t5.dtype}")
    _, _, T, H, W = t5.shape
    exp = {4: 1, 3: W, 2: H * W}
    for d, e in exp.items():
        if t5.shape[d] > 1 and t5.stride(d) != e:
            raise RuntimeError(f"{name}: spatial dims must be dense (stride {t5.stride()})")
    if t5.shape[1] > 1 and t5.stride(1) < T * H * W:
        raise RuntimeError(f"{name}: bad channel stride")


def _bs(t5):
    """(batch stride, channel stride) in elements, tolerant of size-1 dims."""
    _, C, T, H, W = t5.shape
    cs = t5.stride(1) if C > 1 else T * H * W
    bs = t5.stride(0) if t5.shape[0] > 1 else cs * C
    return bs, cs


_GEOM_CACHE = {}


def _geom_arrays(g, x5, y5):
    xb, xc = _bs(x5)
    yb, yc = _bs(y5)
    key = (tuple(g), xb, xc, yb, yc)
    hit = _GEOM_CACHE.get(key)
    if hit is None:
        hit = ((ctypes.c_int * len(g))(*g), (ctypes.c_int64 * 4)(xb, xc, yb, yc))
        _GEOM_CACHE[key] = hit
    return hit


_MAKE_GEOM_CACHE = {}


def _make_geom(B, Cin, Cout, in_thw, k, s, levels, spatial_valid):
    if levels is not None and tuple(k) == (1, 1, 1) and tuple(s) == (1, 1, 1):
        levels = None       # a 1x1 convolution has no tap that could cross a level boundary: the plain (faster) kernels apply
    key = (B, Cin, Cout, tuple(in_thw), k, s, levels if levels is None else tuple(levels), spatial_valid)
    hit = _MAKE_GEOM_CACHE.get(key)
    if hit is None:
        hit = make_geom(B, Cin, Cout, in_thw, k, s, levels, spatial_valid)
        _MAKE_GEOM_CACHE[key] = hit
    return hit


def _opt(t):
    return L.ptr(t) if t is not None else None


def _k3(k):
    return (k, 1, 1) if isinstance(k, int) else tuple(k)



# ----------------------------------------------------------------------------- persistent conv prologues
class PrologueCache:
    """Per-(layer, mode) regions holding what a bf16 conv launch otherwise rebuilds every time in its workspace: the
    gather tables and, for fwd / dgrad, the weights re-packed to bf16 (see include/opental_hip.h, "Persistent
    prologues").  Protocol (DetectorTrainer.step): `activate()` at the START of a step refreshes every known fwd / dgrad
    region from the current weights in ONE launch, then every conv of the step that finds its region skips its own
    prologue; a launch seen for the first time builds its region on the spot (and is part of the batch from then on).
    Regions are only used between activate() and deactivate(), i.e. while the weights cannot change under them."""

    def __init__(self, persistent_range=None):
        # [lo, hi) device address range of the weights that outlive a step (the trainer's flat parameter arena).  Only
        # weights inside it are cached: the batch descriptors keep raw pointers, and a temporary (e.g. the transposed
        # weight of the collapsed projection) would be freed under them.
        self.persistent_range = persistent_range
        self.entries = {}            # key -> region tensor | None (no reusable prologue)
        self.descs = []              # host descriptors (bytes) of the fwd / dgrad regions
        self.blocks = []             # workgroups each of them needs
        self.dev_starts = None
        self.dev_descs = None
        self.dirty = False
        self.pending_join = False    # the refresh runs on the side lane and nobody has waited for it yet

    def region(self, mode, ga, sa, key, w, prec):
        if key in self.entries:
            return self.entries[key]
        lib = L.lib()
        lib.otal_conv_prologue_bytes.restype = ctypes.c_size_t
        lib.otal_conv_prologue_desc_bytes.restype = ctypes.c_size_t
        nbytes = lib.otal_conv_prologue_bytes(ga, sa, mode, prec)
        if nbytes == 0:
            self.entries[key] = None
            return None
        dev = w.device if w is not None else torch.device("cuda", torch.cuda.current_device())
        reg = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        desc = ctypes.create_string_buffer(lib.otal_conv_prologue_desc_bytes())
        rc = lib.otal_conv_prologue(ga, sa, mode, L.ptr(w) if w is not None else None, prec, L.ptr(reg),
                                    ctypes.c_size_t(nbytes), desc, L.stream())
        if rc < 0:
            raise RuntimeError(f"otal_conv_prologue failed: {rc}")
        if rc > 0:                   # fwd / dgrad: needs refreshing whenever the weights change
            self.descs.append(desc.raw)
            self.blocks.append(int(rc))
            self.dirty = True
        self.entries[key] = reg
        return reg

    def refresh(self):
        """Re-pack every registered fwd / dgrad region from the current weights: one launch."""
        if not self.descs:
            return
        if self.dirty:
            host = torch.frombuffer(bytearray(b"".join(self.descs)), dtype=torch.uint8)
            dev = next(iter(v for v in self.entries.values() if v is not None)).device
            self.dev_descs = host.to(dev)
            starts = [0]
            for nb in self.blocks:
                starts.append(starts[-1] + nb)
            self.dev_starts = torch.tensor(starts, dtype=torch.int32).to(dev)
            self.total_blocks = starts[-1]
            self.dirty = False
        def launch():
            L.check(L.lib().otal_conv_prologue_batch(len(self.descs), L.ptr(self.dev_descs), L.ptr(self.dev_starts),
                                                     int(self.total_blocks), L.stream()), "otal_conv_prologue_batch")
        if LANES is not None and PREP_LANE:
            # lane-graph step: the re-pack (0.54 GB of traffic, ~140 us) goes to the SIDE lane and the main lane starts with the
            # first convolution, which packs its own weights (Conv3d_1a's tile kernel, matrix-bound: 380 us + the pool behind
            # it); the main lane waits for the side lane in front of the first launch that reads a region (_prologue)
            LANES.side_chunk(launch)
            self.pending_join = True
        else:
            launch()


PROLOGUES = None        # the active PrologueCache, or None (every launch builds its own prologue)
# OTAL_PREP_LANE=1: in lane graphs the step's weight re-pack runs on the side lane beside Conv3d_1a's forward.  Measured +-0
# (r05f timeline: the pack's 142 us disappear from the main lane and conv1a_tile_fwd_kernel stretches by 110 us): off.
PREP_LANE = os.environ.get("OTAL_PREP_LANE", "0") == "1"


def activate_prologues(cache):
    global PROLOGUES
    PROLOGUES = cache
    if cache is not None:
        cache.refresh()


def deactivate_prologues():
    global PROLOGUES
    c = PROLOGUES
    if c is not None and c.pending_join:         # nobody consumed a region: the side lane is joined all the same
        if LANES is not None:
            LANES.cut(("join",))
        c.pending_join = False
    PROLOGUES = None


def _prologue(mode, ga, sa, pkey, w, prec):
    """pkey: the launch's static identity (mode, geometry, strides) from its plan."""
    c = PROLOGUES
    if c is None or not (prec & 1):
        return None
    wp = 0
    if mode != 2:
        wp = w.data_ptr()
        r = c.persistent_range
        if r is None or not (r[0] <= wp < r[1]):
            return None
    reg = c.region(mode, ga, sa, (wp, pkey, prec), w, prec)
    if reg is None:
        return None
    if c.pending_join and LANES is not None:
        LANES.cut(("join",))
        c.pending_join = False
    return L.ptr(reg)


# ---- launch plans.  Everything about a convolution launch that depends only on shapes and strides -- the geometry record,
# its ctypes arrays, the layout checks, the static part of the prologue key -- is derived once per (mode, shapes, strides,
# kernel, stride, levels) and found again with one dictionary lookup: the wrappers below run ~180 times per training step
# and their Python overhead (about half of a ~17 us call) is what makes eager data-parallel ranks host-bound.
_PLANS = {}


def _plan(mode, xs, ys, Cout, k, s, levels, spatial_valid, xname, yname):
    """xs: the tensor on the convolution's input side (x or dx), ys: on its output side (y or dy); (B,C,T) or (B,C,T,H,W)."""
    key = (mode, xs.shape, xs.stride(), ys.shape, ys.stride(), Cout, k, s, levels, spatial_valid, xs.dtype, ys.dtype,
           xs.device, ys.device)       # the device: _check's is_cuda test must not be skipped by a hit from another device
    plan = _PLANS.get(key)
    if plan is None:
        x5, y5 = _as5(xs), _as5(ys)
        B, Cin, Ti, Hi, Wi = x5.shape
        g, outn = _make_geom(B, Cin, Cout, (Ti, Hi, Wi), k, s, levels, spatial_valid)
        if tuple(y5.shape) != (B, Cout) + outn:
            raise RuntimeError(f"{yname} has shape {tuple(y5.shape)}, expected {(B, Cout) + outn}")
        _check(x5, xname, x5.dtype)         # layout checks; the callers have checked the dtypes
        _check(y5, yname, y5.dtype)
        ga, sa = _geom_arrays(g, x5, y5)
        plan = (g, ga, sa, (mode, tuple(g), _bs(x5), _bs(y5)), (tuple(x5.shape), _bs(x5)))
        if len(_PLANS) > 4096:
            _PLANS.clear()
        _PLANS[key] = plan
    return plan


def _out_positions(x, Cout, k, s, levels, spatial_valid):
    x5 = _as5(x)
    B, Cin, Ti, Hi, Wi = x5.shape
    return _make_geom(B, Cin, Cout, (Ti, Hi, Wi), k, s, levels, spatial_valid)[1]


def half_storage_ok(mode, x_shape, cout, k, s, both=False):
    """True when the library has a kernel for this geometry with the large operand (fwd: y, wgrad: dy) stored as bf16;
    both=True: with the tensors on BOTH sides of the layer stored as bf16 (precision bit 3)."""
    if not (HALF_STORAGE and int(CONV_PRECISION) & 1) or len(x_shape) != 5:
        return False
    k, s = _k3(k), _k3(s)
    B, Cin, Ti, Hi, Wi = x_shape
    g, outn = _make_geom(B, Cin, cout, (Ti, Hi, Wi), k, s, None, False)
    ga = (ctypes.c_int * len(g))(*g)
    P = outn[0] * outn[1] * outn[2]
    sa = (ctypes.c_int64 * 4)(Cin * Ti * Hi * Wi, Ti * Hi * Wi, cout * P, P)
    return bool(L.lib().otal_conv_half_storage(ga, sa, int(mode), int(CONV_PRECISION) | (12 if both else 0)))


def conv_forward(x, w, k, s, scale=None, shift=None, relu=False, spatial_valid=False, levels=None, out=None, half_out=False):
    """y = act(scale * conv_SAME(x, w) + shift).  x (B,Cin,T[,H,W]); w (Cout,Cin,*k).
    half_out: y is STORED as bf16 (half_storage_ok(0, ...) geometries only).  A bfloat16 x (a bf16-STORED activation of the
    backbone, HALF_STORAGE) implies a bfloat16 y: precision bits 2 and 3 of the C ABI."""
    k, s = _k3(k), _k3(s)
    if levels is not None:
        levels = tuple(levels)
    Cout = w.shape[0]
    if x.dtype == torch.bfloat16:
        return _conv_forward_half(x, w, k, s, scale, shift, relu, spatial_valid, levels, out)
    if out is None:
        outn = _out_positions(x, Cout, k, s, levels, spatial_valid)
        out = torch.empty((x.shape[0], Cout) + (outn if x.dim() == 5 else outn[:1]),
                          dtype=torch.bfloat16 if half_out else x.dtype, device=x.device)
    if out.dtype != (torch.bfloat16 if half_out else torch.float32) or x.dtype != torch.float32:
        raise RuntimeError("conv_forward: float32 tensors (bfloat16 y with half_out)")
    g, ga, sa, pkey, _ = _plan(0, x, out, Cout, k, s, levels, spatial_valid, "x", "y")
    if not w.is_contiguous():
        raise RuntimeError("weights must be contiguous")
    wsp, wsn = _ws_args(x.device)
    ev = _prof_begin()
    prec = int(CONV_PRECISION)
    pre = _prologue(0, ga, sa, pkey, w, prec)
    L.check(L.lib().otal_conv_fwd(ga, sa, L.ptr(x), L.ptr(w), _opt(scale), _opt(shift), L.ptr(out), int(relu),
                                  prec | (4 if half_out else 0), pre, wsp, wsn, L.stream()),
            "otal_conv_fwd")
    _prof_end(ev, "fwd", g)
    return out


def _conv_forward_half(x, w, k, s, scale, shift, relu, spatial_valid, levels, out):
    Cout = w.shape[0]
    if out is None:
        outn = _out_positions(x, Cout, k, s, levels, spatial_valid)
        out = torch.empty((x.shape[0], Cout) + (outn if x.dim() == 5 else outn[:1]), dtype=torch.bfloat16, device=x.device)
    if out.dtype != torch.bfloat16 or not (int(CONV_PRECISION) & 1):
        raise RuntimeError("conv_forward: a bfloat16 x needs a bfloat16 y and the bf16-operand mode")
    g, ga, sa, pkey, _ = _plan(0, x, out, Cout, k, s, levels, spatial_valid, "x", "y")
    if not w.is_contiguous():
        raise RuntimeError("weights must be contiguous")
    wsp, wsn = _ws_args(x.device)
    ev = _prof_begin()
    prec = int(CONV_PRECISION)
    pre = _prologue(0, ga, sa, pkey, w, prec)
    L.check(L.lib().otal_conv_fwd(ga, sa, L.ptr(x), L.ptr(w), _opt(scale), _opt(shift), L.ptr(out), int(relu),
                                  prec | 12, pre, wsp, wsn, L.stream()), "otal_conv_fwd (bf16 tensors)")
    _prof_end(ev, "fwd", g)
    return out


def pack_wt(w):
    """(Cout,Cin,*k) -> (Cin,Cout,kvol) packed A operand of the data-gradient GEMM."""
    Cout, Cin = w.shape[0], w.shape[1]
    kvol = w[0, 0].numel()
    wt = torch.empty((Cin, Cout, kvol), dtype=w.dtype, device=w.device)
    L.check(L.lib().otal_conv_pack_wt(L.ptr(w), L.ptr(wt), Cout, Cin, kvol, L.stream()), "otal_conv_pack_wt")
    return wt


def conv_dgrad(dy, w, x_shape, k, s, spatial_valid=False, levels=None, out=None,
               accumulate=False, wt=None, out_mask=None, out_scale=None):
    """dx (+)= d conv / d x.  `out` may be a channel slice.
    out_mask/out_scale: multiply this contribution by (out_mask > 0) * out_scale[ci] in the store
    (ReLU + frozen-BN backward of the layer that produced x; out_mask has dx's layout)."""
    k, s = _k3(k), _k3(s)
    if levels is not None:
        levels = tuple(levels)
    Cout = w.shape[0]
    if out is None:
        if accumulate:
            raise RuntimeError("accumulate needs an existing buffer")
        out = torch.empty(tuple(x_shape), dtype=dy.dtype, device=dy.device)
    elif tuple(out.shape) != tuple(x_shape):
        raise RuntimeError(f"conv_dgrad: out has shape {tuple(out.shape)}, expected {tuple(x_shape)}")
    half = dy.dtype == torch.bfloat16                       # bf16-stored gradients (HALF_STORAGE): dy, dx and the mask alike
    if out.dtype != dy.dtype or not (half or dy.dtype == torch.float32):
        raise RuntimeError("conv_dgrad: float32 tensors, or bfloat16 dy / dx / out_mask together")
    if half and (accumulate or not (int(CONV_PRECISION) & 1)):
        raise RuntimeError("conv_dgrad: bfloat16 tensors need the bf16-operand mode and a plain store")
    g, ga, sa, pkey, xlay = _plan(1, out, dy, Cout, k, s, levels, spatial_valid, "dx", "dy")
    prec = int(CONV_PRECISION)
    if wt is None:          # hand the forward-layout weight over; the launch re-orders it in its own prologue
        wt = w if w.is_contiguous() else w.contiguous()
        prec |= 2
    wsp, wsn = _ws_args(dy.device)
    ev = _prof_begin()
    if out_mask is not None:
        m5 = _as5(out_mask)
        if (tuple(m5.shape), _bs(m5)) != xlay or out_mask.dtype != dy.dtype:
            raise RuntimeError("out_mask must share dx's shape, layout and dtype")
    pre = _prologue(1, ga, sa, pkey, wt, prec) if (prec & 2) else None     # regions are keyed on the live weight tensor
    if half:
        prec |= 12 | (16 if out_mask is not None else 0)
    L.check(L.lib().otal_conv_dgrad(ga, sa, L.ptr(dy), L.ptr(wt), L.ptr(out),
                                    int(accumulate), _opt(out_mask), _opt(out_scale), prec, pre,
                                    wsp, wsn, L.stream()),
            "otal_conv_dgrad")
    _prof_end(ev, "dgrad", g)
    return out


def is_full_collapse(x_shape, k, s, spatial_valid):
    """Unit3D 'spatial_valid' projection whose kernel covers the whole H x W plane (BDNet.py:129-155)."""
    return (spatial_valid and len(x_shape) == 5 and tuple(x_shape[3:]) == tuple(k[1:]) and k[0] == 1
            and tuple(s) == (1, 1, 1))


def conv_dgrad_collapse(dy, w, x_shape):
    """Data gradient of a full-collapse projection as ONE plain GEMM with the operands' roles swapped.

    Every input element (ci, t, h, w) is reached by exactly one tap, so the generic data-gradient
    gather would visit kh*kw taps to find it (36x redundant work for the [1,6,6] projection).
    Instead: dx[(b,t)][(ci,hw)] = sum_co dy[b, co, t] * W[co, (ci,hw)] -- a forward-mode 1x1 GEMM whose
    "activation map" is the WEIGHT, read in place as a (1, Cout, Cin*kh*kw) channel-major tensor (61 MB for the
    [1,6,6] projection: no transposed copy, no per-step bf16 re-pack of it), and whose "weights" are dy transposed to
    [(b,t)][co] (1 MB).  The result is returned as a (B,Cin,T,H,W) VIEW of the [(b,t)][ci][hw] product;
    the backbone's backward un-permutes it inside the launch that applies its ReLU / BatchNorm factor
    (masked_scale_copy), so the map is never copied on its own.
    (Round 1 ran the GEMM on a transposed copy of W: 96 us for the copy, 22 us to re-pack it and 37 us to permute the
    result, around a 49 us GEMM.)"""
    B, Cin, T, H, W = x_shape
    Cout = w.shape[0]
    wa = w.detach().reshape(1, Cout, Cin * H * W)
    dyt = dy.reshape(B, Cout, T).permute(0, 2, 1).reshape(B * T, Cout, 1).contiguous()
    dxp = conv_forward(wa, dyt, 1, 1)                                   # (1, B*T, Cin*H*W)
    return dxp.view(B, T, Cin, H, W).permute(0, 2, 1, 3, 4)


def masked_scale_copy(src, z, scale, dst, accumulate=False):
    """dst (+)= (z > 0) * scale[c] * src for (B,C,T,H,W) tensors with dense H x W planes and otherwise arbitrary strides
    (otal_masked_scale_copy): ReLU + frozen-BN backward, the un-permute of a conv_dgrad_collapse view and the copy into
    a channel slice, in one pass."""
    B, C, T, H, W = dst.shape
    S = H * W
    for name, t in (("src", src), ("z", z), ("dst", dst)):
        if not t.is_cuda or t.dtype != torch.float32 or tuple(t.shape) != (B, C, T, H, W):
            raise RuntimeError(f"masked_scale_copy: {name} must be a float32 GPU tensor of shape {(B, C, T, H, W)}")
        if (W > 1 and t.stride(4) != 1) or (H > 1 and t.stride(3) != W):
            raise RuntimeError(f"masked_scale_copy: {name} needs dense H x W planes (strides {t.stride()})")
    st = lambda t: (ctypes.c_int64 * 3)(t.stride(0), t.stride(1), t.stride(2))
    L.check(L.lib().otal_masked_scale_copy(L.ptr(src), st(src), L.ptr(z), st(z), _opt(scale), L.ptr(dst), st(dst),
                                           int(accumulate), B, C, T, S, L.stream()), "otal_masked_scale_copy")
    return dst


class GradSlots:
    """Address map from a flat parameter arena to its flat gradient arena: while a trainer's backward runs, a weight
    gradient is computed straight into its parameter's slice of the gradient arena instead of a fresh tensor that is
    copied there afterwards (179 MB read + written per step).  A slot is handed out once per step: a weight used twice
    (self-supervised branch) gets a fresh tensor the second time and autograd adds it in."""

    def __init__(self, flat, grad, offsets, numels):
        self.base, self.n, self.grad = flat.data_ptr(), flat.numel(), grad
        self.offsets, self.numels = list(offsets), list(numels)
        self.written = [False] * len(self.offsets)

    def reset(self):
        self.written = [False] * len(self.offsets)

    def take(self, w):
        if w.dtype != torch.float32 or not w.is_contiguous():
            return None
        off = w.data_ptr() - self.base
        if off < 0 or off % 4 or off // 4 + w.numel() > self.n:
            return None
        off //= 4
        i = bisect.bisect_left(self.offsets, off)
        if i == len(self.offsets) or self.offsets[i] != off:
            return None
        j, cur, end = i, off, off + w.numel()          # w may span several adjacent parameters (fused 1x1 weights)
        while j < len(self.offsets) and cur < end:
            cur += self.numels[j]
            j += 1
        if cur != end:
            return None
        if any(self.written[i:j]):
            # second use of a parameter in this backward (a module applied twice): the caller gets a fresh tensor and
            # autograd ADDS the first use's slot to it -- so whatever is still deferred into that slot (GroupNorm batch
            # sums, recorded split-K reductions) has to be written now, not at the next bucket flush
            flush_pending_sums()
            flush_reduces()
            return None
        for q in range(i, j):
            self.written[q] = True
        return self.grad[off:end].view(w.shape)

    def release(self, w):
        """Give a slot back that take(w) handed out and nothing was written to."""
        off = (w.data_ptr() - self.base) // 4
        i = bisect.bisect_left(self.offsets, off)
        j, cur, end = i, off, off + w.numel()
        while j < len(self.offsets) and cur < end:
            self.written[j] = False
            cur += self.numels[j]
            j += 1


GRAD_SLOTS = None       # the running trainer's GradSlots (DetectorTrainer.begin_backward / end_backward)
GRAD_READY = None       # the running trainer's callback for weight gradients that are final before their node returns


def grads_ready(pairs):
    """A multi-layer autograd node (the I3D backbone) announces (weight, gradient) pairs as soon as they are final, so a
    data-parallel trainer can hand the finished arena range to RCCL while the node's remaining layers still run."""
    flush_reduces()                 # the announced gradients are final: their recorded reductions run now, as one launch
    cb = GRAD_READY
    if cb is not None:
        cb(pairs)


def grad_slot(w):
    """Destination of `w`'s gradient in the running trainer's arena, or None (no trainer, foreign weight, slot taken)."""
    return GRAD_SLOTS.take(w) if GRAD_SLOTS is not None and "OTAL_NO_GRAD_SLOTS" not in os.environ else None


def conv_wgrad(x, dy, w_shape, k, s, spatial_valid=False, levels=None, out=None, accumulate=False):
    """dw (+)= d conv / d w."""
    k, s = _k3(k), _k3(s)
    if levels is not None:
        levels = tuple(levels)
    Cout = w_shape[0]
    half_dy = dy.dtype == torch.bfloat16                    # bf16-stored gradient (maxpool3d_backward(half_out=True))
    half_x = x.dtype == torch.bfloat16                      # ... and a bf16-stored activation (HALF_STORAGE): only with a bf16 dy
    if not (half_x or x.dtype == torch.float32) or not (half_dy or dy.dtype == torch.float32) or (half_x and not half_dy):
        raise RuntimeError("conv_wgrad: float32 x with a float32 or bfloat16 dy, or both bfloat16")
    g, ga, sa, pkey, _ = _plan(2, x, dy, Cout, k, s, levels, spatial_valid, "x", "dy")
    if out is None:
        if accumulate:
            raise RuntimeError("accumulate needs an existing buffer")
        out = torch.empty(tuple(w_shape), dtype=torch.float32, device=x.device)
    if not out.is_contiguous():
        raise RuntimeError("dw must be contiguous")
    if _DEFER and (accumulate or (_DEFER_OWNER is not None and _DEFER_OWNER[0] != _WS_SIDE)):
        flush_reduces()             # a recorded reduction may still be on its way to this very buffer / sits on the other stream
    wsp, wsn = _ws_args(x.device)
    ev = _prof_begin()
    prec = int(CONV_PRECISION)
    pre = _prologue(2, ga, sa, pkey, x, prec) if (half_x or not half_dy) else None
    L.check(L.lib().otal_conv_wgrad(ga, sa, L.ptr(x), L.ptr(dy), L.ptr(out),
                                    int(accumulate), prec | (4 if half_dy else 0) | (8 if half_x else 0), pre, wsp, wsn, L.stream()),
            "otal_conv_wgrad")
    if _DEFER:
        _after_wgrad(x.device)
    _prof_end(ev, "wgrad", g)
    return out


# ----------------------------------------------------------------------------- pair launches (two sibling 1-D layers, one grid)
PAIR_LAUNCHES = os.environ.get("OTAL_PAIR_LAUNCHES", "1") != "0"
_VP2 = ctypes.c_void_p * 2


def _pp(a, b):
    return _VP2(a.data_ptr() if a is not None else None, b.data_ptr() if b is not None else None)


def _same_layout(a, b):
    return a.shape == b.shape and a.stride() == b.stride() and a.dtype == b.dtype and a.device == b.device


def conv_forward_pair(xs, ws, k, s, shifts=None, levels=None):
    """(y0, y1) = conv_SAME(x_i, w_i) + shift_i for two problems of one geometry in ONE launch (otal_conv_fwd_pair), or None
    when the library has no pair kernel for it (the caller then runs them one after the other)."""
    k, s = _k3(k), _k3(s)
    if levels is not None:
        levels = tuple(levels)
    if not PAIR_LAUNCHES or not (int(CONV_PRECISION) & 1) or not _same_layout(xs[0], xs[1]) or ws[0].shape != ws[1].shape:
        return None
    if xs[0].dtype != torch.float32 or not (ws[0].is_contiguous() and ws[1].is_contiguous()):
        return None
    Cout = ws[0].shape[0]
    outn = _out_positions(xs[0], Cout, k, s, levels, False)
    ys = [torch.empty((x.shape[0], Cout) + outn[:1], dtype=x.dtype, device=x.device) for x in xs]
    g, ga, sa, pkey, _ = _plan(0, xs[0], ys[0], Cout, k, s, levels, False, "x", "y")
    prec = int(CONV_PRECISION)
    pres = [_prologue(0, ga, sa, pkey, w, prec) for w in ws]
    if (pres[0] is None) != (pres[1] is None):
        return None
    wsp, wsn = _ws_args(xs[0].device)
    ev = _prof_begin()
    sh = None if shifts is None or shifts[0] is None else _pp(shifts[0], shifts[1])
    rc = L.lib().otal_conv_fwd_pair(ga, sa, _pp(*xs), _pp(*ws), None, sh, _pp(*ys), 0, prec,
                                    None if pres[0] is None else _VP2(pres[0].value, pres[1].value), wsp, wsn, L.stream())
    if rc == L.E_UNSUPPORTED:
        return None
    L.check(rc, "otal_conv_fwd_pair")
    _prof_end(ev, "fwd", g, 2)
    return ys


def conv_dgrad_pair(dys, ws, x_shape, k, s, levels=None):
    k, s = _k3(k), _k3(s)
    if levels is not None:
        levels = tuple(levels)
    if not PAIR_LAUNCHES or not (int(CONV_PRECISION) & 1) or not _same_layout(dys[0], dys[1]) or ws[0].shape != ws[1].shape:
        return None
    if dys[0].dtype != torch.float32 or not (ws[0].is_contiguous() and ws[1].is_contiguous()):
        return None
    Cout = ws[0].shape[0]
    outs = [torch.empty(tuple(x_shape), dtype=dy.dtype, device=dy.device) for dy in dys]
    g, ga, sa, pkey, _ = _plan(1, outs[0], dys[0], Cout, k, s, levels, False, "dx", "dy")
    prec = int(CONV_PRECISION) | 2
    pres = [_prologue(1, ga, sa, pkey, w, prec) for w in ws]
    if (pres[0] is None) != (pres[1] is None):
        return None
    wsp, wsn = _ws_args(dys[0].device)
    ev = _prof_begin()
    rc = L.lib().otal_conv_dgrad_pair(ga, sa, _pp(*dys), _pp(*ws), _pp(*outs), prec,
                                      None if pres[0] is None else _VP2(pres[0].value, pres[1].value), wsp, wsn, L.stream())
    if rc == L.E_UNSUPPORTED:
        return None
    L.check(rc, "otal_conv_dgrad_pair")
    _prof_end(ev, "dgrad", g, 2)
    return outs


def conv_wgrad_pair(xs, dys, w_shape, k, s, levels=None, outs=(None, None)):
    k, s = _k3(k), _k3(s)
    if levels is not None:
        levels = tuple(levels)
    if not PAIR_LAUNCHES or not (int(CONV_PRECISION) & 1) or not _same_layout(xs[0], xs[1]) or not _same_layout(dys[0], dys[1]):
        return None
    if xs[0].dtype != torch.float32 or dys[0].dtype != torch.float32:
        return None
    Cout = w_shape[0]
    g, ga, sa, pkey, _ = _plan(2, xs[0], dys[0], Cout, k, s, levels, False, "x", "dy")
    outs = [o if o is not None else torch.empty(tuple(w_shape), dtype=torch.float32, device=xs[0].device) for o in outs]
    if not (outs[0].is_contiguous() and outs[1].is_contiguous()):
        return None
    if _DEFER and _DEFER_OWNER is not None and _DEFER_OWNER[0] != _WS_SIDE:
        flush_reduces()
    wsp, wsn = _ws_args(xs[0].device)
    ev = _prof_begin()
    rc = L.lib().otal_conv_wgrad_pair(ga, sa, _pp(*xs), _pp(*dys), _pp(*outs), int(CONV_PRECISION), wsp, wsn, L.stream())
    if rc == L.E_UNSUPPORTED:
        return None
    L.check(rc, "otal_conv_wgrad_pair")
    if _DEFER:
        _after_wgrad(xs[0].device)
    _prof_end(ev, "wgrad", g, 2)
    return outs


def gn_relu_forward_pair(xs, gammas, betas, groups=32, eps=1e-5, relu=True, levels=None, outs=None):
    """[(y0, stats0), (y1, stats1)] in one launch, or None.  outs: two destinations of ONE layout (see gn_relu_forward)."""
    if not PAIR_LAUNCHES or not _same_layout(xs[0], xs[1]) or not xs[0].is_contiguous():
        return None
    L.require_device(*xs, *gammas, *betas)
    B, C, T = xs[0].shape
    nlev, lev = _lev_arg(levels)
    stats = [torch.empty((B, groups, nlev, 2), dtype=torch.float32, device=x.device) for x in xs]
    if outs is not None:
        st0, st1 = _dest_strides(outs[0], B, C, T), _dest_strides(outs[1], B, C, T)
        if st0 != st1:
            return None
        rc = L.lib().otal_gn_relu_fwd_pair_to(_pp(*xs), _pp(*gammas), _pp(*betas), _pp(*outs), ctypes.c_int64(st0[0]),
                                              ctypes.c_int64(st0[1]), _pp(*stats), B, C, T, groups, ctypes.c_float(eps), int(relu),
                                              nlev, lev, L.stream())
        if rc == L.E_UNSUPPORTED:
            return None
        L.check(rc, "otal_gn_relu_fwd_pair_to")
        return list(zip(outs, stats))
    ys = [torch.empty_like(x) for x in xs]
    rc = L.lib().otal_gn_relu_fwd_pair(_pp(*xs), _pp(*gammas), _pp(*betas), _pp(*ys), _pp(*stats), B, C, T, groups,
                                       ctypes.c_float(eps), int(relu), nlev, lev, L.stream())
    if rc == L.E_UNSUPPORTED:
        return None
    L.check(rc, "otal_gn_relu_fwd_pair")
    return list(zip(ys, stats))


# ----------------------------------------------------------------------------- GroupNorm + ReLU
def _lev_arg(levels):
    if levels is None or len(levels) <= 2:
        return 1, None
    return len(levels) - 1, L.int_array(list(levels))


def _dest_strides(out, B, C, T):
    """(batch stride, channel stride) of a destination that may be a level / channel slice of a wider buffer."""
    if tuple(out.shape) != (B, C, T) or out.dtype != torch.float32 or not out.is_cuda or (T > 1 and out.stride(2) != 1):
        raise RuntimeError("gn_relu_forward: `out` must be a float32 (B,C,T) GPU tensor with dense positions")
    cs = out.stride(1) if C > 1 else T
    return (out.stride(0) if B > 1 else cs * C), cs


def gn_relu_forward(x, gamma, beta, groups=32, eps=1e-5, relu=True, levels=None, out=None):
    """out: write y there -- a level slice of a packed pyramid buffer, a channel slice of a concatenation buffer (the copy
    of torch.cat disappears); its positions must be dense."""
    L.require_device(x, gamma, beta)
    B, C, T = x.shape
    nlev, lev = _lev_arg(levels)
    stats = torch.empty((B, groups, nlev, 2), dtype=torch.float32, device=x.device)
    if out is not None:
        bs, cs = _dest_strides(out, B, C, T)
        L.check(L.lib().otal_gn_relu_fwd_to(L.ptr(x), L.ptr(gamma), L.ptr(beta), L.ptr(out), ctypes.c_int64(bs), ctypes.c_int64(cs),
                                            L.ptr(stats), B, C, T, groups, ctypes.c_float(eps), int(relu), nlev, lev, L.stream()),
                "otal_gn_relu_fwd_to")
        return out, stats
    y = torch.empty_like(x)
    L.check(L.lib().otal_gn_relu_fwd(L.ptr(x), L.ptr(gamma), L.ptr(beta), L.ptr(y), L.ptr(stats), B, C, T, groups,
                                     ctypes.c_float(eps), int(relu), nlev, lev, L.stream()), "otal_gn_relu_fwd")
    return y, stats


def pyramid_merge_forward(p0, p1, total, up):
    """(packed, frame): packed (B,C,total) with level 0 = p0 + nearest-upsampled p1 and level 1 = p1 filled in (the other
    levels are left for their producers), frame (B,C,t0*up) = nearest-upsampled level 0 (BDNet.py:310-326)."""
    L.require_device(p0, p1)
    B, C, t0 = p0.shape
    if tuple(p1.shape) != (B, C, t0 // 2):
        raise RuntimeError("pyramid_merge_forward: p1 must have half of p0's positions")
    packed = torch.empty((B, C, total), dtype=torch.float32, device=p0.device)
    frame = torch.empty((B, C, t0 * up), dtype=torch.float32, device=p0.device)
    L.check(L.lib().otal_pyramid_merge_fwd(L.ptr(p0), L.ptr(p1), L.ptr(packed), L.ptr(frame), B, C, t0, total, up, L.stream()),
            "otal_pyramid_merge_fwd")
    return packed, frame


def pyramid_merge_backward(da, db, dframe, dnext, t0, up):
    """(dp0, dp1) from the gradients w.r.t. the packed buffer (da + db; db may be None), the frame-level input and the
    stride-2 layer's data gradient at level 1 (may be None)."""
    L.require_device(da, dframe)
    B, C, T = da.shape
    dp0 = torch.empty((B, C, t0), dtype=torch.float32, device=da.device)
    dp1 = torch.empty((B, C, t0 // 2), dtype=torch.float32, device=da.device)
    for t in (db, dnext):
        if t is not None:
            L.require_device(t)
    L.check(L.lib().otal_pyramid_merge_bwd(L.ptr(da), _opt(db), L.ptr(dframe), _opt(dnext), L.ptr(dp0), L.ptr(dp1), B, C, t0, T, up,
                                           L.stream()), "otal_pyramid_merge_bwd")
    return dp0, dp1


PENDING_SUMS = None     # the running trainer's list of deferred batch sums (DetectorTrainer.begin_backward), or None


def gn_relu_backward(dy, x, gamma, beta, stats, groups=32, relu=True, levels=None, bias=None):
    """(dx, d_gamma, d_beta, d_conv_bias).  dy may be a channel slice of a wider (B, Ctot, T) map (read in place).
    With a running trainer (PENDING_SUMS) whose gradient arena has free slots for gamma, beta and `bias` (the preceding
    convolution's bias parameter), the three batch sums are DEFERRED: the returned gradients are the arena slices and
    flush_pending_sums() fills them -- one launch for all layers pending at the next bucket flush."""
    B, C, T = x.shape
    if dy.dim() != 3 or tuple(dy.shape) != (B, C, T) or dy.stride(2) != 1 or dy.stride(1) != T or dy.stride(0) < C * T:
        dy = dy.contiguous()
    L.require_device(x, gamma, beta, stats)
    if not dy.is_cuda:
        raise RuntimeError("opental_amd ops run on the GPU only")
    nlev, lev = _lev_arg(levels)
    dx = torch.empty_like(x)
    partial = torch.empty((B, 3, C), dtype=torch.float32, device=x.device)
    L.check(L.lib().otal_gn_relu_bwd(L.ptr(dy), ctypes.c_int64(dy.stride(0)), L.ptr(x), L.ptr(gamma), L.ptr(beta), L.ptr(stats),
                                     L.ptr(dx), L.ptr(partial), B, C, T, groups, int(relu), nlev, lev, L.stream()),
            "otal_gn_relu_bwd")
    return (dx,) + _gn_sums(partial, gamma, beta, bias, C, B)


def _gn_sums(partial, gamma, beta, bias, C, B):
    """(d_gamma, d_beta, d_conv_bias) from the backward kernel's per-sample partials: deferred into the gradient arena when
    a trainer is running (see gn_relu_backward), else summed now."""
    if PENDING_SUMS is not None and bias is not None and GRAD_SLOTS is not None:
        slots = [GRAD_SLOTS.take(t) for t in (gamma, beta, bias)]
        if all(sl is not None for sl in slots):
            # only ADDRESSES are kept: a second reference to a slot tensor would make autograd clone it instead of
            # adopting it as .grad (AccumulateGrad steals a gradient only while it holds the sole reference), and the
            # clone -- taken before the deferred sum has run -- would later be copied over the sum
            PENDING_SUMS.append((partial, tuple(sl.data_ptr() for sl in slots), C, B))
            return slots[0], slots[1], slots[2]
        for t, sl in zip((gamma, beta, bias), slots):       # hand back what was taken: the plain path returns fresh tensors
            if sl is not None:
                GRAD_SLOTS.release(t)
    red = partial.sum(0)            # (3,C): d_gamma, d_beta, d_conv_bias -- three contiguous rows, no copies
    return red[0], red[1], red[2]


def gn_relu_backward_pair(dys, xs, gammas, betas, stats, groups=32, relu=True, levels=None, biases=(None, None)):
    """Two gn_relu_backward calls of one shape in ONE launch: [(dx, d_gamma, d_beta, d_bias)] * 2, or None."""
    if not PAIR_LAUNCHES or not _same_layout(xs[0], xs[1]):
        return None
    B, C, T = xs[0].shape
    dys = list(dys)
    for i in range(2):
        dy = dys[i]
        if dy.dim() != 3 or tuple(dy.shape) != (B, C, T) or dy.stride(2) != 1 or dy.stride(1) != T or dy.stride(0) < C * T:
            dys[i] = dy.contiguous()
        if not dys[i].is_cuda:
            raise RuntimeError("opental_amd ops run on the GPU only")
    L.require_device(*xs, *gammas, *betas, *stats)
    nlev, lev = _lev_arg(levels)
    dxs = [torch.empty_like(x) for x in xs]
    partials = [torch.empty((B, 3, C), dtype=torch.float32, device=x.device) for x in xs]
    rc = L.lib().otal_gn_relu_bwd_pair(_pp(*dys), (ctypes.c_int64 * 2)(dys[0].stride(0), dys[1].stride(0)), _pp(*xs), _pp(*gammas),
                                       _pp(*betas), _pp(*stats), _pp(*dxs), _pp(*partials), B, C, T, groups, int(relu), nlev, lev,
                                       L.stream())
    if rc == L.E_UNSUPPORTED:
        return None
    L.check(rc, "otal_gn_relu_bwd_pair")
    return [(dxs[i],) + _gn_sums(partials[i], gammas[i], betas[i], biases[i], C, B) for i in range(2)]


def flush_pending_sums():
    """Run the deferred batch sums of gn_relu_backward: one launch for every layer pending (otal_sum_partials)."""
    items = PENDING_SUMS
    if not items:
        return
    n = len(items)
    VP = lambda ps: (ctypes.c_void_p * n)(*ps)
    L.check(L.lib().otal_sum_partials(n, VP([it[0].data_ptr() for it in items]), VP([it[1][0] for it in items]),
                                      VP([it[1][1] for it in items]), VP([it[1][2] for it in items]), L.int_array([it[2] for it in items]),
                                      L.int_array([it[3] for it in items]), L.stream()), "otal_sum_partials")
    del items[:]


# ----------------------------------------------------------------------------- MaxPool3d (SAME, zero pad)
def _pool_geom(x5, k, s):
    from .conv_geom import same_pad
    B, C, Ti, Hi, Wi = x5.shape
    pads, outs = [], []
    for size, kk, ss in zip((Ti, Hi, Wi), k, s):
        f, o = same_pad(size, kk, ss)
        pads.append(f)
        outs.append(o)
    return [B, C, Ti, Hi, Wi, *outs, *k, *s, *pads], tuple(outs)


# OTAL_POOL_KEYS=1: the strided pools behind a conv + ReLU take the ordered-key (v_max3_u32) forward kernel.  Bit-identical
# to the scanning kernel and within +-5 % of it either way once both were freed of their serialized loads: off.
POOL_KEYS = os.environ.get("OTAL_POOL_KEYS", "0") == "1"


def maxpool3d_forward(x, k, s, out=None, signbits=False, half_out=False, nonneg=False):
    """(y, winner bytes) -- with signbits=True (y, winner bytes, sign bits of x or None): the strided 3x3 pools can hand
    the ReLU mask of their input to the backward pass as one bit per element (maxpool3d_backward(out_signbits=...)).
    half_out (bfloat16 x only): y is STORED as bf16 too (otal_maxpool3d_fwd_io, io = 3).
    nonneg (with half_out): x is the output of a conv + ReLU, i.e. >= +0 everywhere (io bit 2: the ordered-key kernels)."""
    x5 = _as5(x)
    g, outn = _pool_geom(x5, k, s)
    B, C = x5.shape[:2]
    if half_out:
        if x5.dtype != torch.bfloat16:
            raise RuntimeError("maxpool3d_forward(half_out): a bfloat16 input (winners are copied, not rounded)")
        if out is None:
            out = torch.empty((B, C) + outn, dtype=torch.bfloat16, device=x.device)
        arg = torch.empty((B, C) + outn, dtype=torch.uint8, device=x.device)
        _check(x5, "x", torch.bfloat16); _check(out, "y", torch.bfloat16)
        ga, sa = _geom_arrays(g, x5, out)
        lib = L.lib()
        bits = None
        if signbits:
            lib.otal_maxpool3d_signbits_bytes.restype = ctypes.c_size_t
            nbytes = int(lib.otal_maxpool3d_signbits_bytes(ga, sa))
            if nbytes:
                bits = torch.empty(nbytes, dtype=torch.uint8, device=x.device)
        L.check(lib.otal_maxpool3d_fwd_io(ga, sa, L.ptr(x5), L.ptr(out), L.ptr(arg), _opt(bits), 7 if nonneg else 3, L.stream()),
                "otal_maxpool3d_fwd_io")
        return (out, arg, bits) if signbits else (out, arg)
    if out is None:
        out = torch.empty((B, C) + outn, dtype=x.dtype, device=x.device)
    arg = torch.empty((B, C) + outn, dtype=torch.uint8, device=x.device)
    if x5.dtype == torch.bfloat16:                          # bf16-stored input (conv_forward(half_out=True)): fp32 output
        if out is None or out.dtype != torch.float32:
            out = torch.empty((B, C) + outn, dtype=torch.float32, device=x.device)
        _check(x5, "x", torch.bfloat16); _check(out, "y")
        ga, sa = _geom_arrays(g, x5, out)
        lib = L.lib()
        lib.otal_maxpool3d_signbits_bytes.restype = ctypes.c_size_t
        nbytes = int(lib.otal_maxpool3d_signbits_bytes(ga, sa))
        if not (signbits and nbytes):
            raise RuntimeError("maxpool3d_forward: a bf16 input needs the sign-bit path of the (1,3,3)/(1,2,2) pools")
        bits = torch.empty(nbytes, dtype=torch.uint8, device=x.device)
        L.check(lib.otal_maxpool3d_fwd_signbits_h(ga, sa, L.ptr(x5), L.ptr(out), L.ptr(arg), L.ptr(bits), L.stream()),
                "otal_maxpool3d_fwd_signbits_h")
        return out, arg, bits
    _check(x5, "x"); _check(out, "y")
    ga, sa = _geom_arrays(g, x5, out)
    if signbits:
        lib = L.lib()
        lib.otal_maxpool3d_signbits_bytes.restype = ctypes.c_size_t
        nbytes = int(lib.otal_maxpool3d_signbits_bytes(ga, sa))
        if nbytes and x5.data_ptr() % 16 == 0 and out.data_ptr() % 8 == 0:
            bits = torch.empty(nbytes, dtype=torch.uint8, device=x.device)
            L.check(lib.otal_maxpool3d_fwd_signbits(ga, sa, L.ptr(x5), L.ptr(out), L.ptr(arg), L.ptr(bits), L.stream()),
                    "otal_maxpool3d_fwd_signbits")
            return out, arg, bits
    L.check(L.lib().otal_maxpool3d_fwd(ga, sa, L.ptr(x5), L.ptr(out), L.ptr(arg), L.stream()), "otal_maxpool3d_fwd")
    return (out, arg, None) if signbits else (out, arg)


def maxpool3d_backward(dy, arg, x_shape, k, s, out=None, accumulate=False, out_mask=None, out_scale=None, out_signbits=None,
                       half_out=False):
    """half_out: dx is STORED as bf16 (sign-bit mask, no accumulate) for conv_wgrad's bf16-dy kernels.
    A bfloat16 dy (HALF_STORAGE) implies a bfloat16 dx and, when given, a bfloat16 out_mask (otal_maxpool3d_bwd_io)."""
    if dy.dtype == torch.bfloat16:
        if out is None:
            if accumulate:
                raise RuntimeError("accumulate needs an existing buffer")
            out = torch.empty(tuple(x_shape), dtype=torch.bfloat16, device=dy.device)
        g, outn = _pool_geom(out, k, s)
        _check(out, "dx", torch.bfloat16); _check(dy, "dy", torch.bfloat16)
        if tuple(dy.shape[2:]) != outn or not arg.is_contiguous():
            raise RuntimeError("maxpool3d_backward: shape mismatch")
        if out_mask is not None and (tuple(out_mask.shape) != tuple(out.shape) or out_mask.stride() != out.stride()
                                     or out_mask.dtype != torch.bfloat16):
            raise RuntimeError("out_mask must share dx's shape, layout and dtype")
        ga, sa = _geom_arrays(g, out, dy)
        L.check(L.lib().otal_maxpool3d_bwd_io(ga, sa, L.ptr(dy), L.ptr(arg), L.ptr(out), int(accumulate), _opt(out_mask),
                                              _opt(out_scale), _opt(out_signbits), 3 | (4 if out_mask is not None else 0), L.stream()),
                "otal_maxpool3d_bwd_io")
        return out
    if half_out:
        if accumulate or out_mask is not None or out_signbits is None or out is not None:
            raise RuntimeError("maxpool3d_backward(half_out): plain store with the sign-bit mask only")
        out = torch.empty(tuple(x_shape), dtype=torch.bfloat16, device=dy.device)
        g, outn = _pool_geom(out, k, s)
        _check(out, "dx", torch.bfloat16); _check(dy, "dy")
        if tuple(dy.shape[2:]) != outn or not arg.is_contiguous():
            raise RuntimeError("maxpool3d_backward: shape mismatch")
        ga, sa = _geom_arrays(g, out, dy)
        L.check(L.lib().otal_maxpool3d_bwd_signbits_h(ga, sa, L.ptr(dy), L.ptr(arg), L.ptr(out), L.ptr(out_signbits),
                                                      L.ptr(out_scale), L.stream()), "otal_maxpool3d_bwd_signbits_h")
        return out
    if out is None:
        if accumulate:
            raise RuntimeError("accumulate needs an existing buffer")
        out = torch.empty(tuple(x_shape), dtype=dy.dtype, device=dy.device)
    g, outn = _pool_geom(out, k, s)
    _check(out, "dx"); _check(dy, "dy")
    if tuple(dy.shape[2:]) != outn or not arg.is_contiguous():
        raise RuntimeError("maxpool3d_backward: shape mismatch")
    ga, sa = _geom_arrays(g, out, dy)
    if out_signbits is not None:
        if out.data_ptr() % 16 or dy.data_ptr() % 8:
            raise RuntimeError("maxpool3d_backward: the sign-bit path needs 16-byte aligned dx")
        L.check(L.lib().otal_maxpool3d_bwd_signbits(ga, sa, L.ptr(dy), L.ptr(arg), L.ptr(out), int(accumulate), L.ptr(out_signbits),
                                                    L.ptr(out_scale), L.stream()), "otal_maxpool3d_bwd_signbits")
        return out
    if out_mask is not None and (tuple(out_mask.shape) != tuple(out.shape) or out_mask.stride() != out.stride()):
        raise RuntimeError("out_mask must share dx's shape and layout")
    L.check(L.lib().otal_maxpool3d_bwd(ga, sa, L.ptr(dy), L.ptr(arg), L.ptr(out), int(accumulate),
                                       _opt(out_mask), _opt(out_scale), L.stream()), "otal_maxpool3d_bwd")
    return out


def convert_storage(src, dtype, out=None):
    """A (B,C,T,H,W) / (B,C,T) map (dense positions; may be a channel slice) converted between float32 and bfloat16 STORAGE
    (otal_convert_storage): the boundary between the backbone's bf16-stored tensors and the fp32 tensors around them."""
    s5 = _as5(src)
    if out is None:
        out = torch.empty(tuple(src.shape), dtype=dtype, device=src.device)
    o5 = _as5(out)
    if {src.dtype, out.dtype} != {torch.float32, torch.bfloat16} or tuple(o5.shape) != tuple(s5.shape):
        raise RuntimeError("convert_storage: float32 <-> bfloat16, same shape")
    _check(s5, "src", s5.dtype); _check(o5, "dst", o5.dtype)
    B, C, T, H, W = s5.shape
    if (T * H * W) % 8:
        # the kernel moves eight positions per lane; a map whose position count is not a multiple of eight (short clips, odd
        # T behind a strided pool) takes the elementwise conversion of the tensor library -- the same round-to-nearest-even
        out.copy_(src)
        return out
    (sb, sc), (db, dc) = _bs(s5), _bs(o5)
    L.check(L.lib().otal_convert_storage(L.ptr(s5), ctypes.c_int64(sb), ctypes.c_int64(sc), L.ptr(o5), ctypes.c_int64(db),
                                         ctypes.c_int64(dc), int(out.dtype == torch.bfloat16), B, C, T * H * W, L.stream()),
            "otal_convert_storage")
    return out


# ----------------------------------------------------------------------------- proposal windows / Adam
def proposal_windows(loc, levels, frame_num):
    """loc (B,Ntot,2) -> (segments, frame_segments), both (B,Ntot,4) float32; BDNet.py:355-384."""
    L.require_device(loc)
    B, N, _ = loc.shape
    assert N == levels[-1]
    seg = torch.empty((B, N, 4), dtype=torch.float32, device=loc.device)
    fseg = torch.empty_like(seg)
    L.check(L.lib().otal_proposal_windows(L.ptr(loc), L.ptr(seg), L.ptr(fseg), B, len(levels) - 1,
                                          L.int_array(list(levels)), ctypes.c_float(frame_num), L.stream()),
            "otal_proposal_windows")
    return seg, fseg


def adam_flat(p, g, m, v, step, lr, beta1=0.9, beta2=0.999, eps=1e-8, weight_decay=0.0, grad_scale=1.0):
    L.require_device(p, g, m, v)
    L.check(L.lib().otal_adam_flat(L.ptr(p), L.ptr(g), L.ptr(m), L.ptr(v), ctypes.c_int64(p.numel()),
                                   ctypes.c_float(lr), ctypes.c_float(beta1), ctypes.c_float(beta2),
                                   ctypes.c_float(eps), ctypes.c_float(weight_decay), int(step),
                                   ctypes.c_float(grad_scale), L.stream()), "otal_adam_flat")


def adam_bias_corrections(step, beta1=0.9, beta2=0.999):
    """{1 - beta1^t, sqrt(1 - beta2^t)} in double, rounded once -- the values otal_adam_flat derives on the host."""
    import math
    import numpy as np
    b1, b2 = float(np.float32(beta1)), float(np.float32(beta2))      # the C entry point receives the betas as floats
    return [float(np.float32(1.0 - math.pow(b1, step))), float(np.float32(math.sqrt(1.0 - math.pow(b2, step))))]


def adam_flat_dev(p, g, m, v, bias_corr, lr, beta1=0.9, beta2=0.999, eps=1e-8, weight_decay=0.0, grad_scale=1.0):
    """adam_flat with the bias corrections in a 2-float device tensor (graph-replayable launch)."""
    L.require_device(p, g, m, v, bias_corr)
    L.check(L.lib().otal_adam_flat_dev(L.ptr(p), L.ptr(g), L.ptr(m), L.ptr(v), ctypes.c_int64(p.numel()),
                                       ctypes.c_float(lr), ctypes.c_float(beta1), ctypes.c_float(beta2),
                                       ctypes.c_float(eps), ctypes.c_float(weight_decay), L.ptr(bias_corr),
                                       ctypes.c_float(grad_scale), L.stream()), "otal_adam_flat_dev")


# ----------------------------------------------------------------------------- fused detection loss
class DetectionLossFunction(torch.autograd.Function):
    """The seven loss terms of MultiSegmentLoss (EDL recipe) from ONE launch that also produces every gradient
    (csrc/loss.hip); backward only scales the stored gradients by the incoming scalars."""

    @staticmethod
    def forward(ctx, loc, conf, prop_loc, prop_conf, center, act, prop_act, priors, gt, gvalid, weight_accum,
                clip_length, overlap, ibm_active, num_bins, momentum, iou_aware, cls_mode=0, focal_alpha=0.25):
        B, K, C = conf.shape
        G = gt.shape[1]
        tens = [t.contiguous().float() for t in (loc, conf, prop_loc, prop_conf, center, act, prop_act, priors, gt)]
        L.require_device(*tens)
        gv = gvalid.contiguous().to(torch.uint8)
        lib = L.lib()
        lib.otal_detection_loss_grad_floats.restype = ctypes.c_size_t
        lib.otal_detection_loss_scratch_floats.restype = ctypes.c_size_t
        ng = lib.otal_detection_loss_grad_floats(B, K, C)
        ns = lib.otal_detection_loss_scratch_floats(B, K)
        losses = torch.empty(7, dtype=torch.float32, device=loc.device)
        grads = torch.empty(ng, dtype=torch.float32, device=loc.device)
        scratch = torch.empty(ns, dtype=torch.float32, device=loc.device)
        L.check(lib.otal_detection_loss(*[L.ptr(t) for t in tens], L.ptr(gv), L.ptr(weight_accum), B, K, C, G,
                                        ctypes.c_float(clip_length), ctypes.c_float(overlap), int(ibm_active),
                                        int(num_bins), ctypes.c_float(momentum), int(iou_aware), int(cls_mode),
                                        ctypes.c_float(focal_alpha), L.ptr(losses), L.ptr(grads), L.ptr(scratch), L.stream()),
                "otal_detection_loss")
        ctx.save_for_backward(grads)
        ctx.dims = (B, K, C)
        return tuple(losses[i] for i in range(7))

    @staticmethod
    def backward(ctx, g_l, g_c, g_pl, g_pc, g_ct, g_a, g_pa):
        (grads,) = ctx.saved_tensors
        B, K, C = ctx.dims
        A = B * K
        gs = [None if g is None else g.contiguous() for g in (g_l, g_c, g_pl, g_pc, g_ct, g_a, g_pa)]
        out = torch.empty(4 * A + 2 * A * C + 3 * A, dtype=torch.float32, device=grads.device)
        g7 = (ctypes.c_void_p * 7)(*[None if g is None else g.data_ptr() for g in gs])
        L.check(L.lib().otal_detection_loss_bwd(L.ptr(grads), g7, L.ptr(out), B, K, C, L.stream()), "otal_detection_loss_bwd")
        o = 0
        def take(n, shape):
            nonlocal o
            v = out[o:o + n].view(shape)
            o += n
            return v
        d_loc, d_pl = take(2 * A, (B, K, 2)), take(2 * A, (B, K, 2))
        d_conf, d_pconf = take(A * C, (B, K, C)), take(A * C, (B, K, C))
        d_cen, d_act, d_pact = take(A, (B, K)), take(A, (B, K)), take(A, (B, K))
        return (d_loc, d_conf, d_pl, d_pconf, d_cen, d_act, d_pact) + (None,) * 12


class AnetDetectionLossFunction(torch.autograd.Function):
    """The seven terms of the ActivityNet1.3 MultiSegmentLoss (per-sample normalisation) and all their gradients from one
    workgroup per sample + a seven-sum launch (csrc/loss.hip: otal_detection_loss_anet); backward = the shared
    otal_detection_loss_bwd."""

    @staticmethod
    def forward(ctx, loc, conf, prop_loc, prop_conf, center, act, prop_act, priors2, gt, gvalid, level_bounds,
                clip_length, overlap, ibm_active, ibm_coeff, iou_aware, act_weight, act_margin):
        B, K, C = conf.shape
        G = gt.shape[1]
        tens = [t.contiguous().float() for t in (loc, conf, prop_loc, prop_conf, center, act, prop_act, priors2, gt)]
        L.require_device(*tens)
        gv = gvalid.contiguous().to(torch.uint8)
        lib = L.lib()
        lib.otal_detection_loss_grad_floats.restype = ctypes.c_size_t
        ng = lib.otal_detection_loss_grad_floats(B, K, C)
        losses = torch.empty(7, dtype=torch.float32, device=loc.device)
        grads = torch.empty(ng, dtype=torch.float32, device=loc.device)
        scratch = torch.empty(8 * B, dtype=torch.float32, device=loc.device)
        nlev = len(level_bounds)
        lbs = (ctypes.c_float * (2 * nlev))(*[float(v) for row in level_bounds for v in row])
        L.check(lib.otal_detection_loss_anet(*[L.ptr(t) for t in tens], L.ptr(gv), B, K, C, G, ctypes.c_float(clip_length),
                                             ctypes.c_float(overlap), lbs, nlev, int(ibm_active), ctypes.c_float(ibm_coeff),
                                             int(iou_aware), ctypes.c_float(act_weight), ctypes.c_float(act_margin),
                                             L.ptr(losses), L.ptr(grads), L.ptr(scratch), L.stream()), "otal_detection_loss_anet")
        ctx.save_for_backward(grads)
        ctx.dims = (B, K, C)
        return tuple(losses[i] for i in range(7))

    @staticmethod
    def backward(ctx, *gs):
        return DetectionLossFunction.backward(ctx, *gs)[:7] + (None,) * 11


# ----------------------------------------------------------------------------- head output tails
HEAD_WGRAD_SIDE = os.environ.get("OTAL_HEAD_WGRAD_SIDE", "1") != "0"     # the fused heads' weight gradients on the weight-gradient lane
HEAD_GRAD_SLOTS = "OTAL_NO_GRAD_SLOTS_HEADS" not in os.environ      # (A/B switch of the head bias / ScaleExp gradient slots)


_SLOT_INDEX = {}        # arena offsets of a group of one-element gradient slots, as a device index tensor


def _adjacent_view(ts):
    """One (n,) view over n one-element fp32 tensors that lie next to each other in ONE storage, in order; else None."""
    if not ts or any(t is None or t.numel() != 1 or t.dtype != torch.float32 for t in ts):
        return None
    base = ts[0]
    st = base.untyped_storage().data_ptr()
    if any(t.untyped_storage().data_ptr() != st or t.data_ptr() != base.data_ptr() + 4 * i for i, t in enumerate(ts)):
        return None
    return base.detach().as_strided((len(ts),), (1,), base.storage_offset())


class HeadOutputsFunction(torch.autograd.Function):
    """permute / ScaleExp / Dirichlet uncertainty of all detection-head maps in one launch (csrc/heads.hip).

    apply(levels, level_strides, modes, *scales_then_raws): the nlev one-element ScaleExp parameters (separate tensors, so
    that each receives its own 16-byte-aligned gradient: a 4-byte-offset view would push the trainer's whole multi-tensor
    gradient copy onto its unvectorised path), then raws[i] (B,C_i,N) -> outs[i] (B,N,C_i) for every item, followed by
    the uncertainty maps (B,N) of the mode-2 items."""

    @staticmethod
    def forward(ctx, levels, level_strides, modes, *tensors):
        nlev_ = len(levels) - 1
        ctx.scale_params = tensors[:nlev_]
        scales = _adjacent_view(tensors[:nlev_])             # the parameters where they lie (the trainer's arena), or a packed copy
        if scales is None:
            scales = torch.cat([t.detach().reshape(1) for t in tensors[:nlev_]])
        raws = [r.contiguous() for r in tensors[nlev_:]]
        L.require_device(scales, *raws)
        B, _, N = raws[0].shape
        n = len(raws)
        outs = [torch.empty((B, N, r.shape[1]), dtype=torch.float32, device=r.device) for r in raws]
        uncts = [torch.empty((B, N), dtype=torch.float32, device=raws[0].device) if m == 2 else None for m in modes]
        VP = ctypes.c_void_p * n
        arr = lambda ts: VP(*[None if t is None else t.data_ptr() for t in ts])
        strides = None if level_strides is None else (ctypes.c_float * len(level_strides))(*level_strides)
        meta = (L.int_array([r.shape[1] for r in raws]), L.int_array(list(modes)), len(levels) - 1, L.int_array(list(levels)), strides, B, N, n)
        L.check(L.lib().otal_head_outputs_fwd(n, meta[0], meta[1], arr(raws), arr(outs), arr(uncts), L.ptr(scales), B, N, meta[2],
                                              meta[3], strides, L.stream()), "otal_head_outputs_fwd")
        ctx.meta, ctx.modes = meta, tuple(modes)
        ctx.save_for_backward(scales, *raws, *outs, *[u for u in uncts if u is not None])
        return tuple(outs) + tuple(u for u in uncts if u is not None)

    @staticmethod
    def backward(ctx, *grads):
        chans, modes_a, nlev, lev, strides, B, N, n = ctx.meta
        saved = ctx.saved_tensors
        scales, raws, outs, us = saved[0], saved[1:1 + n], saved[1 + n:1 + 2 * n], list(saved[1 + 2 * n:])
        uncts, dunct, k = [], [], 0
        for m in ctx.modes:
            if m == 2:
                uncts.append(us[k]); dunct.append(grads[n + k]); k += 1
            else:
                uncts.append(None); dunct.append(None)
        douts = [None if g is None else g.contiguous() for g in grads[:n]]
        dunct = [None if g is None else g.contiguous() for g in dunct]
        draws = [torch.empty_like(r) for r in raws]
        VP = ctypes.c_void_p * n
        arr = lambda ts: VP(*[None if t is None else t.data_ptr() for t in ts])

        def launch(dscales):
            L.check(L.lib().otal_head_outputs_bwd(n, chans, modes_a, arr(raws), arr(outs), arr(uncts), arr(douts), arr(dunct), arr(draws),
                                                  L.ptr(scales), L.ptr(dscales), B, N, nlev, lev, strides, L.stream()),
                    "otal_head_outputs_bwd")
        # the nlev one-element gradients: written in place when the parameters' arena slots are adjacent (no packing, no copies:
        # they were 6 of the 13 tiny gradient copies -- a hipMemcpyAsync node each -- of every step's bucket flushes)
        want = HEAD_GRAD_SLOTS and all(ctx.needs_input_grad[3:3 + nlev])     # (the refined stage passes the scales detached)
        slots = [grad_slot(p) if want else None for p in ctx.scale_params]
        direct = _adjacent_view(slots) if all(sl is not None for sl in slots) else None
        if direct is not None:
            direct.zero_()
            launch(direct)
            return (None, None, None) + tuple(slots) + tuple(draws)
        dscales = torch.zeros_like(scales)
        launch(dscales)
        if all(sl is not None for sl in slots):     # slots anywhere in the arena (e.g. in descending order): ONE scatter launch
            key = tuple(sl.data_ptr() for sl in slots)
            idx = _SLOT_INDEX.get(key)
            if idx is None:
                flat = GRAD_SLOTS.grad
                idx = torch.tensor([(sl.data_ptr() - flat.data_ptr()) // 4 for sl in slots], dtype=torch.int64).to(flat.device)
                if len(_SLOT_INDEX) > 64:
                    _SLOT_INDEX.clear()
                _SLOT_INDEX[key] = idx
            GRAD_SLOTS.grad.index_copy_(0, idx, dscales)
            return (None, None, None) + tuple(slots) + tuple(draws)
        for p, sl in zip(ctx.scale_params, slots):
            if sl is not None:
                GRAD_SLOTS.release(p)
        aligned = torch.zeros((nlev, 4), dtype=dscales.dtype, device=dscales.device)
        aligned[:, 0] = dscales
        return (None, None, None) + tuple(aligned[l, :1] for l in range(nlev)) + tuple(draws)


# ----------------------------------------------------------------------------- detection-head convolutions
def _hc_meta(heads, n_inputs):
    n = len(heads)
    return (n, n_inputs, L.int_array([h[0] for h in heads]), L.int_array([h[1] for h in heads]), L.int_array([h[2] for h in heads]))


def head_convs_supported(heads, n_inputs, B, C, N):
    """heads: [(input index, cout, k)].  True when the heads of a stage fit the fused launches (csrc/headconv.hip)."""
    if not 1 <= len(heads) <= 8 or not 1 <= n_inputs <= 4:
        return False
    n, ni, idx, co, ks = _hc_meta(heads, n_inputs)
    return bool(L.lib().otal_head_convs_supported(n, ni, idx, co, ks, int(B), int(C), int(N)))


class HeadConvsFunction(torch.autograd.Function):
    """The Unit1D heads of one CoarsePyramid stage (BDNet.py:337-353 / :399-412) -- conv1d(512, cout, k) + bias with
    cout = 1 / 2 / num_classes -- in one launch forward and two backward (otal_head_convs_fwd / _bwd).

    apply(levels, heads, n_inputs, *tensors): heads = ((input index, k), ...); tensors = the n_inputs feature maps
    (B, C, N), then every head's weight (cout, C, k), then every head's bias.  Returns the raw maps (B, cout, N)."""

    @staticmethod
    def forward(ctx, levels, heads, n_inputs, *tensors):
        nh = len(heads)
        xs = [t.contiguous() for t in tensors[:n_inputs]]
        ws = list(tensors[n_inputs:n_inputs + nh])
        bs = list(tensors[n_inputs + nh:n_inputs + 2 * nh])
        L.require_device(*xs, *ws, *[b for b in bs if b is not None])
        B, C, N = xs[0].shape
        spec = [(heads[i][0], int(ws[i].shape[0]), int(heads[i][1])) for i in range(nh)]
        for (j, co, k), w in zip(spec, ws):
            if tuple(w.shape) != (co, C, k) or tuple(xs[j].shape) != (B, C, N) or w.dtype != torch.float32:
                raise RuntimeError("head_convs: weight / input shapes do not match")
        meta = _hc_meta(spec, n_inputs)
        nlev, lev = _lev_arg(levels)
        ys = [torch.empty((B, co, N), dtype=torch.float32, device=xs[0].device) for _, co, _ in spec]
        VP = lambda ts: (ctypes.c_void_p * len(ts))(*[None if t is None else t.data_ptr() for t in ts])
        L.check(L.lib().otal_head_convs_fwd(*meta, VP(xs), VP(ws), VP(bs), VP(ys), B, C, N, nlev, lev, L.stream()),
                "otal_head_convs_fwd")
        ctx.cfg = (meta, nlev, lev, n_inputs, nh, (B, C, N), [b is not None for b in bs])
        ctx.bias_params = bs                # (leaf parameters: only their arena slots are looked up in backward)
        ctx.save_for_backward(*xs, *ws)
        return tuple(ys)

    @staticmethod
    def backward(ctx, *dys):
        meta, nlev, lev, n_inputs, nh, (B, C, N), has_bias = ctx.cfg
        saved = ctx.saved_tensors
        xs, ws = list(saved[:n_inputs]), list(saved[n_inputs:])
        dys = [None if g is None else g.contiguous() for g in dys]
        dxs = [torch.empty_like(x) if ctx.needs_input_grad[3 + j] else None for j, x in enumerate(xs)]
        dws, slots_w, slots_b = [], [], []
        for w in ws:
            slot = grad_slot(w)
            slots_w.append(slot)
            dws.append(slot if slot is not None else torch.empty_like(w))
        # bias gradients go straight to their arena slots too (they were 7 of the 13 one-to-fifteen-element gradient copies --
        # a hipMemcpyAsync node each -- that every step's bucket flushes issued)
        dbs = []
        for w, b in zip(ws, ctx.bias_params):
            slot = grad_slot(b) if (b is not None and HEAD_GRAD_SLOTS) else None
            slots_b.append(slot)
            dbs.append(slot if slot is not None else (torch.empty(w.shape[0], dtype=torch.float32, device=w.device) if b is not None else None))
        VP = lambda ts: (ctypes.c_void_p * len(ts))(*[None if t is None else t.data_ptr() for t in ts])
        in_slots = all(s_ is not None for s_ in slots_w) and all(s_ is not None for s_, b in zip(slots_b, ctx.bias_params) if b is not None)
        side = side_wgrads(xs[0].device)
        if HEAD_WGRAD_SIDE and side.on and in_slots and SIDE_DEFER_JOIN:
            # the data gradients on the chain that waits for them, the weight / bias gradients on the weight-gradient lane (they
            # land in arena slots nobody reads before the trainer's join): 29 + 39 us off the serial middle of the step
            L.check(L.lib().otal_head_convs_bwd_parts(*meta, VP(xs), VP(ws), VP(dys), VP(dxs), VP(dws), VP(dbs), B, C, N, nlev, lev, 1,
                                                      L.stream()), "otal_head_convs_bwd_parts")
            dst_w, dst_b = [t.detach() for t in dws], [None if t is None else t.detach() for t in dbs]
            side.pending.append(lambda: L.check(L.lib().otal_head_convs_bwd_parts(
                *meta, VP(xs), VP(ws), VP(dys), VP([None] * len(xs)), VP(dst_w), VP(dst_b), B, C, N, nlev, lev, 2, L.stream()),
                "otal_head_convs_bwd_parts"))
            side.keep.append((xs, dys, ws))
            side.node_end(True)
        else:
            L.check(L.lib().otal_head_convs_bwd(*meta, VP(xs), VP(ws), VP(dys), VP(dxs), VP(dws), VP(dbs), B, C, N, nlev, lev, L.stream()),
                    "otal_head_convs_bwd")
        return (None, None, None) + tuple(dxs) + tuple(dws) + tuple(dbs)


def head_convs(levels, items):
    """items: [(feature map (B,C,N), Unit1D head)] -> raw maps, or None when the fused launches do not cover the stage
    (e.g. the 150-class ActivityNet heads): the caller then runs the heads one by one."""
    maps, heads = [], []
    for x, unit in items:
        j = next((i for i, m in enumerate(maps) if m is x), None)
        if j is None:
            j = len(maps)
            maps.append(x)
        conv = unit.conv1d
        k = conv.kernel_size[0]
        if conv.stride[0] != 1 or k not in (1, 3) or x.dim() != 3 or x.dtype != torch.float32 or not x.is_cuda:
            return None
        heads.append((j, k, conv))
    B, C, N = maps[0].shape
    if any(tuple(m.shape) != (B, C, N) for m in maps) or C % 16:
        return None
    if not head_convs_supported([(j, conv.out_channels, k) for j, k, conv in heads], len(maps), B, C, N):
        return None
    return HeadConvsFunction.apply(None if levels is None else tuple(levels), tuple((j, k) for j, k, _ in heads), len(maps),
                                   *maps, *[c.weight for _, _, c in heads], *[c.bias for _, _, c in heads])


HC_MAX_ROWS = 21        # csrc/headconv.hip: output channels of all fused heads on one input map


def head_convs_mixed(levels, items, run_one):
    """The heads of a stage where some are too wide for the fused launches (the 150-class heads of the ActivityNet model):
    the skinny ones (1 / 2 output channels) still share ONE forward and TWO backward launches, only the wide ones run as
    convolutions of their own (`run_one(x, unit)`).  Returns the raw maps in the order of `items`, or None when not even the
    skinny subset fits (the caller then runs every head by itself)."""
    wide = [unit.conv1d.out_channels > HC_MAX_ROWS for _, unit in items]
    if not any(wide) or all(wide):
        return None
    small = [it for it, w in zip(items, wide) if not w]
    fused = head_convs(levels, small)
    if fused is None:
        return None
    fused = list(fused)
    return [run_one(x, unit) if w else fused.pop(0) for (x, unit), w in zip(items, wide)]


# ----------------------------------------------------------------------------- boundary (start / end) losses
class BoundaryBCEFunction(torch.autograd.Function):
    """(loss_start, loss_end) = calc_bce_loss on the two channel halves of x (B,C,T), read in place (x may be a slice along
    T of a larger map); mask (B,R,Tm) holds the start / end rows at row0, row0+1 and is sampled every `step` frames."""

    @staticmethod
    def forward(ctx, x, mask, row0, step):
        if not (x.is_cuda and mask.is_cuda):
            raise RuntimeError("opental_amd ops run on the GPU only")
        B, C, T = x.shape
        if x.stride(2) != 1 or mask.stride(2) != 1 or x.dtype != torch.float32 or mask.dtype != torch.float32:
            raise RuntimeError("boundary_bce: float32 maps with unit time stride")
        if row0 + 2 > mask.shape[1] or (T - 1) * step >= mask.shape[2]:
            raise RuntimeError("boundary_bce: mask rows / length do not cover the map")
        terms = torch.empty((B, 2, T), dtype=torch.float32, device=x.device)
        dx = torch.empty((B, C, T), dtype=torch.float32, device=x.device)
        L.check(L.lib().otal_boundary_bce(L.ptr(x), ctypes.c_int64(x.stride(0)), ctypes.c_int64(x.stride(1)), L.ptr(mask),
                                          ctypes.c_int64(mask.stride(0)), ctypes.c_int64(mask.stride(1)), int(row0), int(step),
                                          L.ptr(terms), L.ptr(dx), B, C, T, L.stream()), "otal_boundary_bce")
        ctx.save_for_backward(dx)
        ctx.x_shape = tuple(x.shape)
        losses = terms.sum(dim=(0, 2)) / float(B * T)
        return losses[0], losses[1]

    @staticmethod
    def backward(ctx, g_start, g_end):
        (dx,) = ctx.saved_tensors
        B, C, T = ctx.x_shape
        z = dx.new_zeros(())
        g = torch.stack([z if g_start is None else g_start.reshape(()), z if g_end is None else g_end.reshape(())])
        return (dx.view(B, 2, C // 2, T) * g.view(1, 2, 1, 1)).view(B, C, T), None, None, None


class BoundaryLossesFunction(torch.autograd.Function):
    """(loss_start, loss_end) of a training step: sum_i weights[i] * calc_bce_loss(map_i) over the frame-level map and
    the two level-0 proposal maps (train.py:193-201, weights 1, 0.1, 0.1) -- one otal_boundary_bce launch per map, ONE
    launch for all the means and the weighted sums, ONE for the backward of all maps.
    apply(mask, steps, weights, *maps): maps[i] (B, C_i, T_i) read in place, mask sampled every steps[i] frames."""

    @staticmethod
    def forward(ctx, mask, steps, weights, *xs):
        n = len(xs)
        B = xs[0].shape[0]
        terms, dxs = [], []
        for x, step in zip(xs, steps):
            if not (x.is_cuda and mask.is_cuda) or x.stride(2) != 1 or mask.stride(2) != 1 or x.dtype != torch.float32:
                raise RuntimeError("boundary losses: float32 GPU maps with unit time stride")
            _, C, T = x.shape
            if 2 > mask.shape[1] or (T - 1) * step >= mask.shape[2]:
                raise RuntimeError("boundary losses: mask rows / length do not cover the map")
            t = torch.empty((B, 2, T), dtype=torch.float32, device=x.device)
            dx = torch.empty((B, C, T), dtype=torch.float32, device=x.device)
            L.check(L.lib().otal_boundary_bce(L.ptr(x), ctypes.c_int64(x.stride(0)), ctypes.c_int64(x.stride(1)), L.ptr(mask),
                                              ctypes.c_int64(mask.stride(0)), ctypes.c_int64(mask.stride(1)), 0, int(step),
                                              L.ptr(t), L.ptr(dx), B, C, T, L.stream()), "otal_boundary_bce")
            terms.append(t); dxs.append(dx)
        out = torch.empty(2, dtype=torch.float32, device=xs[0].device)
        wa = (ctypes.c_float * n)(*[float(w) for w in weights])
        VP = lambda ts: (ctypes.c_void_p * n)(*[t.data_ptr() for t in ts])
        L.check(L.lib().otal_boundary_finish(n, VP(terms), wa, L.int_array([t.shape[2] for t in terms]), B, L.ptr(out), L.stream()),
                "otal_boundary_finish")
        ctx.save_for_backward(*dxs)
        ctx.meta = (n, B, [float(w) for w in weights])
        return out[0], out[1]

    @staticmethod
    def backward(ctx, g_start, g_end):
        dxs = ctx.saved_tensors
        n, B, weights = ctx.meta
        outs = [torch.empty_like(d) for d in dxs]
        wa = (ctypes.c_float * n)(*weights)
        VP = lambda ts: (ctypes.c_void_p * n)(*[t.data_ptr() for t in ts])
        gs = [None if g is None else g.contiguous() for g in (g_start, g_end)]
        L.check(L.lib().otal_boundary_scale(n, VP(dxs), VP(outs), wa, L.int_array([d.shape[1] for d in dxs]),
                                            L.int_array([d.shape[2] for d in dxs]), B, _opt(gs[0]), _opt(gs[1]), L.stream()),
                "otal_boundary_scale")
        return (None, None, None) + tuple(outs)
