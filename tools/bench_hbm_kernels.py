"""Achieved HBM GB/s of the bandwidth-bound kernel classes of the detection path, at the shapes of the b=8 THUMOS14
training step (BASELINE north_star: "achieved HBM GB/s for the pooling and 1D-conv kernels ... against gfx950 peak").

achieved = ALGORITHMIC bytes of one launch (SURVEY.md 8d formulas, stated per kernel below) / average launch duration,
measured with HIP events on the launch stream around the replay of a HIP graph holding `reps` back-to-back launches.  Peak: 8 TB/s (MI355X_MICROARCH.md).
All of these working sets (<= 40 MB except the 3-D max-pools and Adam) are resident in the 256 MB Infinity Cache after
the first launch, so for them the figure is a cache-bandwidth / launch-latency statement, flagged `resident`.

    python tools/bench_hbm_kernels.py [batch]          # one JSON object on stdout
"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

PEAK_GBS = 8000.0
LEVELS = (0, 64, 96, 112, 120, 124, 126)


EAGER = None        # counter passes (tools/pmc_hbm_kernels.sh): run every kernel EAGER times with plain launches, no timing


def timed(fn, reps=200, warm=5):
    """Average duration of one launch: `reps` launches are captured in a HIP graph and its replay is timed with HIP
    events, so that the host's launch rate (~10 us per ctypes call) does not hide the microsecond kernels."""
    if EAGER:
        for _ in range(EAGER):
            fn()
        torch.cuda.synchronize()
        return 1.0
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(warm):
            fn()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        for _ in range(reps):
            fn()
    graph.replay()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    a.record()
    graph.replay()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e-3 / reps


def windows(rs, B, lens, T_of):
    """Plausible proposal windows [l_s, r_s, l_e, r_e] around each anchor, in the coordinate system of length T_of(t)."""
    rows = []
    for t in lens:
        T = T_of(t)
        c = (np.arange(t) + 0.5) / t * T
        half = rs.uniform(1.0, T / 4.0, size=(B, t))
        l, r = c[None] - half, c[None] + half
        o, i = np.maximum(2 * half / 10, 1), np.maximum(2 * half / 4, 1)
        rows.append(np.round(np.stack([l - o, l + i, r - i, r + o], -1)))
    return torch.from_numpy(np.concatenate(rows, 1).astype(np.float32)).cuda()


def measure(batch=8):
    from opental_amd.common import ops
    from opental_amd.prop_pooling import boundary_pooling_op as bp
    dev = torch.device("cuda", 0)
    rs = np.random.RandomState(0)
    B = batch
    lens = [LEVELS[i + 1] - LEVELS[i] for i in range(6)]
    N = LEVELS[-1]
    out = {}

    def add(name, seconds, nbytes, resident, note, kernel=None):
        # kernel: substring(s, '|'-separated alternatives) of the device kernel's name (the counter passes attribute FETCH_SIZE / WRITE_SIZE rows by it)
        out[name] = {"us": round(seconds * 1e6, 2), "algorithmic_MB": round(nbytes / 1e6, 3),
                     "GB/s": round(nbytes / seconds / 1e9, 1), "frac_of_8TBps": round(nbytes / seconds / 1e9 / PEAK_GBS, 4),
                     "resident_in_infinity_cache": resident, "what": note, "kernel": kernel}

    # BoundaryMaxPooling, level-batched (lr_conv features C=1024, T=N=126 packed levels): SURVEY 8d
    #   fwd 4*(C*T + 4*N + C*N) per sample, bwd 4*(C*N + C*T + 4*N + C*T)
    C = 1024
    x = torch.randn(B, C, N, device=dev)
    seg = torch.cat([windows(rs, B, [t], lambda t: t) for t in lens], 1)
    g = torch.randn(B, C, N, device=dev)
    add("bmp_levels_fwd", timed(lambda: bp.bmp_forward_levels(x, seg, LEVELS, LEVELS)), 4 * B * (C * N + 4 * N + C * N), True,
        "otal_bmp_fwd_levels, (8,1024,126) features, 126 proposals, one launch for the six levels", "bmp_fwd_kernel")
    add("bmp_levels_bwd", timed(lambda: bp.bmp_backward_levels(g, x, seg, LEVELS, LEVELS)), 4 * B * (C * N + 2 * C * N + 4 * N), True,
        "otal_bmp_bwd_levels (deterministic gather)", "bmp_bwd_kernel")
    # frame-level pooling: C=512, T=256, all 126 proposals in one launch
    C, T = 512, 256
    xf = torch.randn(B, C, T, device=dev)
    fseg = windows(rs, B, lens, lambda t: T)
    gf = torch.randn(B, C, N, device=dev)
    add("bmp_frame_fwd", timed(lambda: bp.bmp_forward(xf, fseg)), 4 * B * (C * T + 4 * N + C * N), True,
        "otal_bmp_fwd, (8,512,256) frame-level features, 126 proposals", "bmp_fwd_kernel")
    add("bmp_frame_bwd", timed(lambda: bp.bmp_backward(gf, xf, fseg)), 4 * B * (C * N + 2 * C * T + 4 * N), True, "otal_bmp_bwd", "bmp_bwd_kernel")

    # Conv1d + GroupNorm + ReLU block of the towers: 512 -> 512, k=3, (8,512,126) level-packed.
    #   conv bytes 4*(Cout*Cin*k + Cout + Cin*t*b + Cout*t*b); GN+ReLU reads and writes the map once
    Cc = 512
    xa = torch.randn(B, Cc, N, device=dev)
    w = torch.randn(Cc, Cc, 3, device=dev) * 0.02
    # (timed as in the training step: the layer's tables and bf16-packed weights live in a persistent prologue region that
    #  is refreshed ONCE per step, not per launch -- round 1 timed each launch together with its own prologue kernels, which
    #  is how the driver saw 56 us where the step pays 24)
    w5 = w.view(Cc, Cc, 3, 1, 1)
    cache = ops.PrologueCache((w5.data_ptr(), w5.data_ptr() + 4 * w5.numel()))
    for prec, tag in ((0, "f32"), (1, "bf16")):
        ops.CONV_PRECISION = prec
        ops.activate_prologues(cache)
        try:
            ops.conv_forward(xa, w5, (3, 1, 1), (1, 1, 1), levels=LEVELS)          # builds the region (first use)
            ops.activate_prologues(cache)                                          # uploads its descriptor (outside any capture)
            add(f"conv1d_k3_512_fwd_{tag}", timed(lambda: ops.conv_forward(xa, w5, (3, 1, 1), (1, 1, 1), levels=LEVELS)),
                4 * (Cc * Cc * 3 + Cc + 2 * Cc * N * B), True,
                f"otal_conv_fwd ({tag} MFMA operands) on the level-packed tower map, persistent prologue as in the step; "
                "weight-read bound (3.1 MB weights vs 2.1 MB activations)", "conv1d_tile_kernel" if prec else "conv_gemm_kernel")
        finally:
            ops.deactivate_prologues()
    ops.CONV_PRECISION = 1
    gamma, beta = torch.ones(Cc, device=dev), torch.zeros(Cc, device=dev)
    add("gn_relu_fwd", timed(lambda: ops.gn_relu_forward(xa, gamma, beta, levels=LEVELS)), 4 * 2 * B * Cc * N, True,
        "otal_gn_relu_fwd, per-level statistics, (8,512,126)", "gn_relu_fwd_kernel")
    y, stats = ops.gn_relu_forward(xa, gamma, beta, levels=LEVELS)
    add("gn_relu_bwd", timed(lambda: ops.gn_relu_backward(xa, xa, gamma, beta, stats, levels=LEVELS)), 4 * 3 * B * Cc * N, True,
        "otal_gn_relu_bwd (+ the (B,C,3) partial sums reduced by torch)", "gn_relu_bwd_kernel")

    # MaxPool3d_2a_3x3: (8,64,128,48,48) -> (.,.,128,24,24), k (1,3,3) s (1,2,2), as the step runs it since round 4: bf16-STORED
    # input and output (ops.HALF_CHAIN).  fwd reads x (2 B), writes y (2 B) + 1-byte tap + the ReLU sign bits (1 bit per input);
    # bwd reads dy (2 B) + tap + sign bits, writes dx (2 B)
    xp = torch.relu(torch.randn(B, 64, 128, 48, 48, device=dev)).to(torch.bfloat16)
    n_in, n_out = xp.numel(), xp.numel() // 4
    yp, arg, bits = ops.maxpool3d_forward(xp, (1, 3, 3), (1, 2, 2), signbits=True, half_out=True)
    add("maxpool_2a_fwd", timed(lambda: ops.maxpool3d_forward(xp, (1, 3, 3), (1, 2, 2), out=yp, signbits=True, half_out=True), reps=20),
        2 * n_in + 3 * n_out + n_in // 8, False, "otal_maxpool3d_fwd_io on Conv3d_1a's bf16-stored output (302 MB in)", "maxpool133_s2_w8_fwd_kernel|maxpoolk33_s2_fwd_kernel")
    dxp = torch.empty_like(xp)
    sc = torch.ones(64, device=dev)
    add("maxpool_2a_bwd", timed(lambda: ops.maxpool3d_backward(yp, arg, xp.shape, (1, 3, 3), (1, 2, 2), out=dxp, out_scale=sc, out_signbits=bits), reps=20),
        3 * n_out + n_in // 8 + 2 * n_in, False, "otal_maxpool3d_bwd_io with the producer's ReLU mask (sign bits) and BN scale fused into the store",
        "maxpool133_s2_w8_bwd_kernel|maxpoolk33_s2_bwd_kernel")
    del xp, yp, arg, dxp, bits
    # the fused 1x1x1 launch of Mixed_3c on bf16-stored tensors (256 -> 288 channels on 8 x 128 x 12 x 12 positions): the HBM-bound
    # convolution class of the backbone.  Bytes: x + y (2 B each) + the packed bf16 weights; the data gradient also reads the mask
    xc = torch.relu(torch.randn(B, 256, 128, 12, 12, device=dev)).to(torch.bfloat16)
    wc = torch.randn(288, 256, 1, 1, 1, device=dev) * 0.05
    scc = torch.ones(288, device=dev)
    yc = ops.conv_forward(xc, wc, (1, 1, 1), (1, 1, 1), scale=scc, shift=scc, relu=True)
    add("conv1x1_3c_fwd_bf16", timed(lambda: ops.conv_forward(xc, wc, (1, 1, 1), (1, 1, 1), scale=scc, shift=scc, relu=True, out=yc), reps=50),
        2 * (xc.numel() + yc.numel()) + 2 * wc.numel(), False, "otal_conv_fwd, bf16 tensors on both sides (precision bits 2 + 3)", "conv1x1_stream_kernel|conv_gemm_bf16c_kernel")
    dxc = torch.empty_like(xc)
    sci = torch.ones(256, device=dev)
    add("conv1x1_3c_dgrad_bf16", timed(lambda: ops.conv_dgrad(yc, wc, xc.shape, (1, 1, 1), (1, 1, 1), out=dxc, out_mask=xc, out_scale=sci), reps=50),
        2 * (2 * xc.numel() + yc.numel()) + 2 * wc.numel(), False, "otal_conv_dgrad with the bf16 activation as ReLU mask", "conv1x1_stream_kernel|conv_gemm_bf16c_kernel")
    del xc, yc, dxc
    # Adam over the flat arena: reads p, g, m, v; writes p, m, v -> 28 B / parameter
    n = 44_720_000
    p, gr, m, v = (torch.randn(n, device=dev) * 0.01 for _ in range(4))
    v.abs_()
    add("adam_flat", timed(lambda: ops.adam_flat(p, gr, m, v, 3, 1e-5, weight_decay=1e-3), reps=20), 28 * n, False,
        "otal_adam_flat over 44.72 M parameters (one launch)", "adam_flat_kernel")
    return out


if __name__ == "__main__":
    if "--eager" in sys.argv:       # counter passes: `--eager N` plain launches per entry; prints the entries in launch order
        EAGER = int(sys.argv[sys.argv.index("--eager") + 1])
    print(json.dumps(measure(int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 8), indent=1))
