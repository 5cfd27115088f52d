"""The fused 1-D block launches (csrc/block1d.hip, otal_b1d_*) against a float64 torch restatement of what they replace:
Unit1D (AFSD/common/layers.py:178-214) + nn.GroupNorm(32, C) + nn.ReLU of AFSD/thumos14/BDNet.py:67-103,:129-203,:274-284
going forward, and that block's GroupNorm / ReLU backward fed by its consumers' data gradients going backward.  Operands
are bf16-valued (the kernel rounds activations and weights to bf16 while staging, as every otal_conv_* launch of the bf16
mode does), accumulation is fp32: tolerance 2e-4 of the output scale, stated per assertion."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

LEV = (0, 64, 96, 112, 120, 124, 126)


def _dev():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    return torch.device("cuda", 0)


def bf(t):
    return t.to(torch.bfloat16).to(torch.float32)


def rnd(rs, *shape, scale=1.0):
    return bf(torch.from_numpy((rs.randn(*shape) * scale).astype(np.float32)))


def same_conv_levels(x, w, lev, stride=1):
    """SAME-padded conv1d applied level by level (float64)."""
    k = w.shape[2]
    outs = []
    for a, b in zip(lev[:-1], lev[1:]):
        xl = x[:, :, a:b]
        if stride == 1:
            outs.append(F.conv1d(F.pad(xl, ((k - 1) // 2, k - 1 - (k - 1) // 2)), w))
        else:       # AFSD/common/layers.py:198-210: even length, stride 2, k = 3 pads (0, 1)
            outs.append(F.conv1d(F.pad(xl, (0, 1)), w, stride=2))
    return torch.cat(outs, 2)


def gn_levels(c, gamma, beta, lev, groups=32, eps=1e-5):
    return torch.cat([F.group_norm(c[:, :, a:b], groups, gamma, beta, eps) for a, b in zip(lev[:-1], lev[1:])], 2)


def close(got, want, tol, what):
    want = want.to(torch.float32)
    err = (got.cpu() - want).abs().max().item()
    scale = want.abs().max().item()
    assert err <= tol * max(scale, 1e-6), (what, err, scale)


@pytest.mark.parametrize("case", ["towers_k3", "lr_k1_cpg32", "stride2", "upsampled", "cat3"])
def test_fused_forward_block(case):
    from opental_amd.common import block1d as B1
    dev = _dev()
    rs = np.random.RandomState(3)
    B = 2
    if case == "towers_k3":
        cin, cout, kt, T, lev, ranges = 512, 512, 3, 126, LEV, (0, 1, 6)
    elif case == "lr_k1_cpg32":
        cin, cout, kt, T, lev, ranges = 512, 1024, 1, 126, LEV, (0, 1, 6)
    elif case == "stride2":
        cin, cout, kt, T, lev, ranges = 512, 512, 3, 16, (0, 16), None
    elif case == "upsampled":
        cin, cout, kt, T, lev, ranges = 512, 512, 3, 256, (0, 256), None
    else:
        cin, cout, kt, T, lev, ranges = 2048, 512, 1, 126, LEV, (0, 1, 6)
    w = rnd(rs, cout, cin, kt, scale=0.05)
    bias, gamma, beta = rnd(rs, cout, scale=0.1), rnd(rs, cout) * 0.5 + 1.0, rnd(rs, cout, scale=0.2)
    if case == "stride2":
        x = rnd(rs, B, cin, 32)
        c_ref = same_conv_levels(x.double(), w.double(), (0, 32), stride=2) + bias.double()[None, :, None]
    elif case == "upsampled":
        x = rnd(rs, B, cin, 64)
        xu = x.double().repeat_interleave(4, dim=2)     # F.interpolate(..., [256, 1]) nearest (BDNet.py:324-325)
        c_ref = same_conv_levels(xu, w.double(), (0, 256)) + bias.double()[None, :, None]
    else:
        x = rnd(rs, B, cin, T)
        c_ref = same_conv_levels(x.double(), w.double(), lev) + bias.double()[None, :, None]
    y_ref = F.relu(gn_levels(c_ref, gamma.double(), beta.double(), lev))

    wd, xd = w.to(dev), x.to(dev)
    pack = B1.Pack(wd, fwd=True, dgrad=False)
    B1.PackSet([pack]).refresh()
    c = torch.empty((B, cout, T), device=dev)
    y = torch.full((B, cout, T), float("nan"), device=dev)
    stats = torch.empty((B, 32, len(lev) - 1, 2), device=dev)
    if case == "stride2":
        segs = [B1.seg(xd, pack.fwd, cin, 3, mul=2, off=0, Tv=32)]
    elif case == "upsampled":
        segs = [B1.seg(xd, pack.fwd, cin, 3, mul=1, off=-1, shr=2, Tv=256)]
    elif case == "cat3":        # torch.cat([roi, pooled, short], 1) as three K segments of one weight (BDNet.py:111-112)
        parts = [xd[:, :512].contiguous(), xd[:, 512:1536].contiguous(), xd[:, 1536:].contiguous()]
        segs, col = [], 0
        for p in parts:
            s, keep = B1.seg(p, pack.fwd, p.shape[1], 1, use_levels=True)
            s.wp = pack.fwd.data_ptr() + col * 2
            s.wp_elems = pack.fwd.numel() - col
            segs.append((s, keep))
            col += p.shape[1]
    else:
        segs = [B1.seg(xd, pack.fwd, cin, kt, off=-(kt // 2), use_levels=True)]
    P = B1.problem(B1.FWD, B, cout, T, segs, y, c=c, stats=stats, gamma=gamma.to(dev), beta=beta.to(dev), bias=bias.to(dev),
                   levels=lev, ranges=ranges)
    assert B1.launch([P])
    torch.cuda.synchronize()
    close(c, c_ref, 2e-4, "conv output")
    close(y, y_ref, 5e-4, "block output")
    # statistics: {mean, rstd} per (sample, group, level)
    cg = c_ref.view(B, 32, cout // 32, T)
    for l, (a, b) in enumerate(zip(lev[:-1], lev[1:])):
        m = cg[:, :, :, a:b].mean((2, 3))
        v = cg[:, :, :, a:b].var((2, 3), unbiased=False)
        close(stats[:, :, l, 0], m, 1e-3, "mean")
        close(stats[:, :, l, 1], 1.0 / torch.sqrt(v + 1e-5), 1e-3, "rstd")


def test_two_problems_in_one_launch_equal_two_launches():
    from opental_amd.common import block1d as B1
    dev = _dev()
    rs = np.random.RandomState(5)
    B, C, T = 2, 512, 126
    outs = []
    ws = [rnd(rs, C, C, 3, scale=0.05).to(dev) for _ in range(2)]
    xs = [rnd(rs, B, C, T).to(dev) for _ in range(2)]
    g, be = torch.ones(C, device=dev), torch.zeros(C, device=dev)
    packs = [B1.Pack(w, dgrad=False) for w in ws]
    B1.PackSet(packs).refresh()

    def prob(i):
        c, y = torch.empty((B, C, T), device=dev), torch.empty((B, C, T), device=dev)
        st = torch.empty((B, 32, 6, 2), device=dev)
        P = B1.problem(B1.FWD, B, C, T, [B1.seg(xs[i], packs[i].fwd, C, 3, off=-1, use_levels=True)], y, c=c, stats=st,
                       gamma=g, beta=be, levels=LEV, ranges=(0, 1, 6))
        return P, (c, y, st)
    a0, a1 = prob(0), prob(1)
    assert B1.launch([a0[0], a1[0]])
    b0, b1 = prob(0), prob(1)
    assert B1.launch([b0[0]]) and B1.launch([b1[0]])
    torch.cuda.synchronize()
    for (_, ta), (_, tb) in ((a0, b0), (a1, b1)):
        for u, v in zip(ta, tb):
            assert torch.equal(u, v)


@pytest.mark.parametrize("case", ["two_consumers_k3", "stride2_consumer", "adds_only", "plain"])
def test_fused_backward_block(case):
    """dc = d cost / d (conv output) of a block whose output y = relu(GN(c)) is read by stride-1 / stride-2 convolutions
    (their output gradients dz are given) and by other consumers whose gradient w.r.t. y arrives as fp32 `adds`."""
    from opental_amd.common import block1d as B1
    dev = _dev()
    rs = np.random.RandomState(7)
    B, C = 2, 512
    if case == "stride2_consumer":
        T, lev, ranges = 32, (0, 32), None
    else:
        T, lev, ranges = 126, LEV, (0, 1, 6)
    c = rnd(rs, B, C, T)
    gamma, beta = rnd(rs, C) * 0.5 + 1.0, rnd(rs, C, scale=0.2)
    cd = c.double().requires_grad_(True)
    gd, bd = gamma.double().requires_grad_(True), beta.double().requires_grad_(True)
    y = F.relu(gn_levels(cd, gd, bd, lev))
    cost = 0
    segs, adds, keep = [], [], []
    if case in ("two_consumers_k3", "plain"):
        for kt in (3, 1):
            w, dz = rnd(rs, C, C, kt, scale=0.05), rnd(rs, B, C, T)
            cost = cost + (same_conv_levels(y if case != "plain" else cd, w.double(), lev) * dz.double()).sum()
            pk = B1.Pack(w.to(dev), fwd=False)
            keep.append(pk)
            segs.append(B1.seg(dz.to(dev), pk.dgrad, C, kt, off=kt // 2, sgn=-1, use_levels=True))
    if case == "stride2_consumer":
        w, dz = rnd(rs, C, C, 3, scale=0.05), rnd(rs, B, C, 16)
        cost = cost + (same_conv_levels(y, w.double(), lev, stride=2) * dz.double()).sum()
        pk = B1.Pack(w.to(dev), fwd=False)
        keep.append(pk)
        segs.append(B1.seg(dz.to(dev), pk.dgrad, C, 3, off=0, sgn=-1, shr=1, par=1, Tv=32))
    e_full, e_lvl0 = rnd(rs, B, C, T), rnd(rs, B, C, 64)
    if case != "stride2_consumer":
        src = y if case != "plain" else cd
        cost = cost + (src * e_full.double()).sum() + (src[:, :, :64] * e_lvl0.double()).sum()
        adds = [(e_full.to(dev), None), (e_lvl0.to(dev), 64)]
    cost.backward()
    if keep:
        B1.PackSet(keep).refresh()
    out = torch.full((B, C, T), float("nan"), device=dev)
    if case == "plain":
        P = B1.problem(B1.PLAIN, B, C, T, segs, out, adds=adds, levels=lev, ranges=ranges)
        assert B1.launch([P])
        torch.cuda.synchronize()
        close(out, cd.grad, 3e-4, "plain data gradient")
        return
    if case == "adds_only":
        segs = []
    # the forward statistics, as the forward launch stores them
    stats = torch.empty((B, 32, len(lev) - 1, 2))
    cg = c.double().view(B, 32, C // 32, T)
    for l, (a, b) in enumerate(zip(lev[:-1], lev[1:])):
        stats[:, :, l, 0] = cg[:, :, :, a:b].mean((2, 3))
        stats[:, :, l, 1] = 1.0 / torch.sqrt(cg[:, :, :, a:b].var((2, 3), unbiased=False) + 1e-5)
    nr = 1 if ranges is None else len(ranges) - 1
    partial = torch.full((B * nr, 3, C), float("nan"), device=dev)
    P = B1.problem(B1.BWD, B, C, T, segs, out, c=c.to(dev), stats=stats.to(dev), gamma=gamma.to(dev), beta=beta.to(dev),
                   adds=adds, partial=partial, levels=lev, ranges=ranges)
    assert B1.launch([P])
    torch.cuda.synchronize()
    close(out, cd.grad, 5e-4, "dc")
    sums = partial.cpu().double().sum(0)
    close(sums[0].float(), gd.grad, 5e-4, "d gamma")
    close(sums[1].float(), bd.grad, 5e-4, "d beta")
    close(sums[2].float(), cd.grad.sum((0, 2)), 2e-3, "d bias")
