"""Geometry vector handed to otal_conv_* (include/opental_hip.h): 19 ints + the level table.

Layout: B,Cin,Cout, Ti,Hi,Wi, To,Ho,Wo, kt,kh,kw, st,sh,sw, pt,ph,pw, nlev, lev[0..8].
SAME padding follows the reference's rule (AFSD/common/layers.py:198-210,
AFSD/common/i3d_backbone.py:46-72): total = max(k - s, 0) if size % s == 0 else max(k - size % s, 0),
front = total // 2; only the front pad is needed (out-of-range taps read zero).
"""
MAX_LEVELS = 8


def same_pad(size, k, s):
    total = max(k - s, 0) if size % s == 0 else max(k - (size % s), 0)
    front = total // 2
    return front, (size + total - k) // s + 1


def make_geom(B, Cin, Cout, in_thw, k, s, levels=None, spatial_valid=False):
    """-> (list[int] geometry, (To,Ho,Wo)).  `spatial_valid`: pad the temporal axis only
    (reference Unit3D padding='spatial_valid', layers.py:161-168)."""
    pads, outs = [], []
    for d in range(3):
        if spatial_valid and d > 0:
            pads.append(0)
            outs.append((in_thw[d] - k[d]) // s[d] + 1)
        else:
            f, o = same_pad(in_thw[d], k[d], s[d])
            pads.append(f)
            outs.append(o)
    nlev = 1
    lev = [0] * (MAX_LEVELS + 1)
    if levels is not None and len(levels) > 2:
        nlev = len(levels) - 1
        assert nlev <= MAX_LEVELS and in_thw[1] == in_thw[2] == 1 and s[0] == 1
        for i in range(MAX_LEVELS + 1):
            lev[i] = levels[min(i, nlev)]
    else:
        lev[1:] = [in_thw[0]] * MAX_LEVELS
    g = [B, Cin, Cout, *in_thw, *outs, *k, *s, *pads, nlev, *lev]
    return g, tuple(outs)
