"""Soft-NMS on MI355X with the reference's signature (AFSD/common/segment_utils.py:128-162).

`softnms_v2(segments, ...)` takes the same (N,3..5) tensor [start, end, score, (unct), (act)] and
returns (rows, count[, done_mask]) with the reference's semantics (first-maximum greedy order, last
survivor never kept, rows in original index order, decayed scores).  The reference moves the data
to the CPU and loops in Python; here the whole greedy loop runs in one workgroup with the
candidates resident in LDS (otal_softnms_classes).  For batches of videos use
`opental_amd.thumos14.test.detect_batch`, which gathers + suppresses all (video, class) pairs in one launch.
"""
import ctypes

import torch

from .. import _lib as L


def softnms_v2(segments, sigma=0.5, top_k=1000, score_threshold=0.001, use_edl=False, os_head=False, get_mask=False):
    if not segments.is_cuda:
        raise RuntimeError("softnms_v2: GPU tensor expected (no CPU fallback in opental_amd)")
    seg = segments.float().contiguous()
    n, cols = seg.shape
    ncol = 3 + int(bool(use_edl)) + int(bool(os_head))
    if cols < ncol:
        raise RuntimeError(f"segments has {cols} columns, {ncol} expected")
    dev = seg.device
    if n == 0:
        empty = seg.new_zeros((0, ncol))
        return (empty, torch.zeros((), dtype=torch.long, device=dev)) + ((torch.zeros(0, dtype=torch.bool, device=dev),) if get_mask else ())
    # one "video" with one "clip" of n anchors and one class
    se = seg[:, :2].contiguous()
    sc = seg[:, 2].contiguous()
    zeros = torch.zeros(n, device=dev)
    c3 = seg[:, 3].contiguous() if cols > 3 else zeros
    c4 = seg[:, 4].contiguous() if cols > 4 else zeros
    flag = torch.ones(n, dtype=torch.uint8, device=dev)
    clip_start = torch.tensor([0, 1], dtype=torch.int32, device=dev)
    k = min(int(top_k), n)
    out = torch.empty((1, max(k, 1), 5), device=dev)
    counts = torch.zeros(1, dtype=torch.int32, device=dev)
    idx = torch.empty((1, max(k, 1)), dtype=torch.int32, device=dev)
    lib = L.lib()
    lib.otal_softnms_scratch_bytes.restype = ctypes.c_size_t
    nbytes = int(lib.otal_softnms_scratch_bytes(1, 1, n, 1))          # > 0 past ~7600 candidates: global working set
    scratch = torch.empty(max(nbytes, 1), dtype=torch.uint8, device=dev)
    L.check(lib.otal_softnms_classes_ws(L.ptr(se), L.ptr(sc), L.ptr(c3), L.ptr(c4), L.ptr(flag), L.ptr(clip_start),
                                        1, 1, n, 1, ctypes.c_float(sigma), max(k, 1), ctypes.c_float(score_threshold),
                                        L.ptr(out), L.ptr(counts), L.ptr(idx), 5, L.ptr(scratch), ctypes.c_size_t(nbytes), 1,
                                        L.stream()), "otal_softnms_classes_ws")
    count = int(counts.item())
    rows = out[0, :count]
    if ncol == 3:
        rows = rows[:, :3]
    elif ncol == 4:
        rows = rows[:, [0, 1, 2, 3 if use_edl else 4]]
    res = (rows, counts[0].long())
    if get_mask:
        mask = torch.zeros(n, dtype=torch.bool, device=dev)
        mask[idx[0, :count].long()] = True
        res = res + (mask,)
    return res
