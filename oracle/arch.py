"""oracle/arch.py -- TEST INFRASTRUCTURE.  Architecture tables + seeded parameter fill.

The reference ships no weights in the container (models/ is a dangling symlink), so golden
vectors are produced from parameters generated here from ``numpy.random.RandomState`` (stable
across machines) in the reference's own state-dict key order.  The key/shape list below was
checked entry by entry against ``BDNet(training=False, use_edl=True).state_dict()`` of the
reference (446 entries, THUMOS14) by ``oracle/pin_against_reference.py``.

Reference: AFSD/common/i3d_backbone.py:194-296 (endpoint table), AFSD/thumos14/BDNet.py:116-293
(CoarsePyramid layers), AFSD/thumos14/BDNet.py:460-473 (glorot-uniform init limits).
"""
import numpy as np

# (name, kind, args) in forward order.  conv: (cin, cout, k, stride); pool: (k, stride);
# mixed: (cin, [b0, b1a, b1b, b2a, b2b, b3b])
I3D_ENDPOINTS = [
    ("Conv3d_1a_7x7", "conv", (None, 64, (7, 7, 7), (2, 2, 2))),
    ("MaxPool3d_2a_3x3", "pool", ((1, 3, 3), (1, 2, 2))),
    ("Conv3d_2b_1x1", "conv", (64, 64, (1, 1, 1), (1, 1, 1))),
    ("Conv3d_2c_3x3", "conv", (64, 192, (3, 3, 3), (1, 1, 1))),
    ("MaxPool3d_3a_3x3", "pool", ((1, 3, 3), (1, 2, 2))),
    ("Mixed_3b", "mixed", (192, (64, 96, 128, 16, 32, 32))),
    ("Mixed_3c", "mixed", (256, (128, 128, 192, 32, 96, 64))),
    ("MaxPool3d_4a_3x3", "pool", ((3, 3, 3), (2, 2, 2))),
    ("Mixed_4b", "mixed", (480, (192, 96, 208, 16, 48, 64))),
    ("Mixed_4c", "mixed", (512, (160, 112, 224, 24, 64, 64))),
    ("Mixed_4d", "mixed", (512, (128, 128, 256, 24, 64, 64))),
    ("Mixed_4e", "mixed", (512, (112, 144, 288, 32, 64, 64))),
    ("Mixed_4f", "mixed", (528, (256, 160, 320, 32, 128, 128))),
    ("MaxPool3d_5a_2x2", "pool", ((2, 2, 2), (2, 2, 2))),
    ("Mixed_5b", "mixed", (832, (256, 160, 320, 32, 128, 128))),
    ("Mixed_5c", "mixed", (832, (384, 192, 384, 48, 128, 128))),
]

MIXED_BRANCH_CONVS = [  # (sub-name, cin source, cout index, kernel)
    ("b0", "in", 0, (1, 1, 1)),
    ("b1a", "in", 1, (1, 1, 1)),
    ("b1b", 1, 2, (3, 3, 3)),
    ("b2a", "in", 3, (1, 1, 1)),
    ("b2b", 3, 4, (3, 3, 3)),
    ("b3b", "in", 5, (1, 1, 1)),
]

THUMOS = dict(name="thumos14", frame_num=256, feat_t=64, layer_num=6, num_classes=15,
              feat_channels=(832, 1024), conv_channels=512, two_projections=True,
              proj_kernels=((1, 6, 6), (1, 3, 3)), fpn_strides=None, loc_bias=(0.5, 3.5))

# ActivityNet1.3 variant (configs/anet_opental.yaml, AFSD/anet/BDNet.py:13-33,:120-269): 768 frames, ONE
# projection (Mixed_5c, [1,3,3]) to t = 96, five stride-2 levels after it, loc in frames = exp(.) * fpn_stride,
# priors carry the level id in a second column, per-level regression bounds (anet/multisegment_loss.py:69).
ANET = dict(name="anet", frame_num=768, feat_t=96, layer_num=6, num_classes=150,
            feat_channels=(1024,), conv_channels=512, two_projections=False,
            proj_kernels=((1, 3, 3),), fpn_strides=(4, 8, 16, 32, 64, 128), loc_bias=(0.0, 1.5),
            bounds=((0, 30), (15, 60), (30, 120), (60, 240), (96, 768), (256, 768)))


def level_lengths(cfg=THUMOS):
    t, out = cfg["feat_t"], []
    for _ in range(cfg["layer_num"]):
        out.append(t)
        t //= 2
    return out


def backbone_conv_list(in_channels=3):
    """[(state-dict prefix, cin, cout, kernel, stride)] for the 50 backbone convs (+BN each)."""
    out = []
    for name, kind, args in I3D_ENDPOINTS:
        if kind == "conv":
            cin, cout, k, s = args
            out.append((f"backbone._model.{name}", in_channels if cin is None else cin, cout, k, s))
        elif kind == "mixed":
            cin, oc = args
            for sub, src, oi, k in MIXED_BRANCH_CONVS:
                c_in = cin if src == "in" else oc[src]
                out.append((f"backbone._model.{name}.{sub}", c_in, oc[oi], k, (1, 1, 1)))
    return out


def _unit1d_gn(prefix, cin, cout, k, conv_idx=0, gn_idx=1):
    return [(f"{prefix}.{conv_idx}.conv1d.weight", (cout, cin, k)),
            (f"{prefix}.{conv_idx}.conv1d.bias", (cout,)),
            (f"{prefix}.{gn_idx}.weight", (cout,)),
            (f"{prefix}.{gn_idx}.bias", (cout,))]


def param_spec(cfg=THUMOS, in_channels=3):
    """Full ordered (key, shape) list == reference state_dict() order (THUMOS14 BDNet)."""
    P = "coarse_pyramid_detection"
    C = cfg["conv_channels"]
    K = cfg["num_classes"]
    spec = []
    for i, (fc, kk) in enumerate(zip(cfg["feat_channels"], cfg["proj_kernels"])):
        spec += [(f"{P}.pyramids.{i}.0.conv3d.weight", (C, fc) + tuple(kk)),
                 (f"{P}.pyramids.{i}.0.conv3d.bias", (C,)),
                 (f"{P}.pyramids.{i}.1.weight", (C,)), (f"{P}.pyramids.{i}.1.bias", (C,))]
    for i in range(len(cfg["proj_kernels"]), cfg["layer_num"]):
        spec += _unit1d_gn(f"{P}.pyramids.{i}", C, C, 3)
    for i in range(cfg["layer_num"]):
        spec.append((f"{P}.loc_heads.{i}.scale", (1,)))
    for tower in ("loc_tower", "conf_tower"):
        for i in range(2):
            spec += _unit1d_gn(f"{P}.{tower}.{i}", C, C, 3)
    for head, co, k in (("loc_head", 2, 3), ("conf_head", K, 3), ("actionness_head", 1, 3)):
        spec += [(f"{P}.{head}.conv1d.weight", (co, C, k)), (f"{P}.{head}.conv1d.bias", (co,))]
    for br in ("loc_proposal_branch", "conf_proposal_branch"):
        spec += _unit1d_gn(f"{P}.{br}.cur_point_conv", C, C, 1)
        spec += _unit1d_gn(f"{P}.{br}.lr_conv", C, 2 * C, 1)
        spec += _unit1d_gn(f"{P}.{br}.roi_conv", C, C, 1)
        spec += _unit1d_gn(f"{P}.{br}.proposal_conv", 4 * C, C, 1)
    for head, co, k in (("prop_loc_head", 2, 1), ("prop_conf_head", K, 1),
                        ("prop_actionness_head", 1, 1), ("center_head", 1, 3)):
        spec += [(f"{P}.{head}.conv1d.weight", (co, C, k)), (f"{P}.{head}.conv1d.bias", (co,))]
    spec += _unit1d_gn(f"{P}.deconv", C, C, 3, 0, 1)
    spec += _unit1d_gn(f"{P}.deconv", C, C, 3, 3, 4)
    spec += _unit1d_gn(f"{P}.deconv", C, C, 1, 6, 7)
    for prefix, cin, cout, k, _s in backbone_conv_list(in_channels):
        spec += [(f"{prefix}.conv3d.weight", (cout, cin) + tuple(k)),
                 (f"{prefix}.bn.weight", (cout,)), (f"{prefix}.bn.bias", (cout,)),
                 (f"{prefix}.bn.running_mean", (cout,)), (f"{prefix}.bn.running_var", (cout,)),
                 (f"{prefix}.bn.num_batches_tracked", ())]
    return spec


def make_params(seed=2020, cfg=THUMOS, in_channels=3, randomize_affine=True, gain=1.0):
    """Seeded numpy parameters (dict key -> np.ndarray) in state-dict order.

    Conv weights: glorot-uniform with the limits of BDNet.weight_init
    (AFSD/thumos14/BDNet.py:460-473): limit = sqrt(3 / max(1, (fan_in + fan_out) / 2)),
    times ``gain`` (gain > 1 keeps activations from vanishing through 60 ReLU layers).
    With ``randomize_affine`` biases / norm affines / BN statistics are drawn at random too,
    so that a mis-indexed bias or scale cannot hide behind the 0 / 1 defaults.
    """
    rs = np.random.RandomState(seed)
    out = {}
    for key, shape in param_spec(cfg, in_channels):
        if key.endswith("num_batches_tracked"):
            out[key] = np.zeros((), np.int64)
        elif key.endswith("conv3d.weight") or key.endswith("conv1d.weight"):
            rf = int(np.prod(shape[2:]))
            fan_in, fan_out = shape[1] * rf, shape[0] * rf
            limit = gain * np.sqrt(3.0 / max(1.0, (fan_in + fan_out) / 2.0))
            out[key] = rs.uniform(-limit, limit, size=shape).astype(np.float32)
        elif key.endswith("running_var"):
            out[key] = (rs.uniform(0.5, 1.5, size=shape) if randomize_affine
                        else np.ones(shape)).astype(np.float32)
        elif key.endswith("running_mean"):
            out[key] = (rs.uniform(-0.1, 0.1, size=shape) if randomize_affine
                        else np.zeros(shape)).astype(np.float32)
        elif key.endswith(".scale"):
            out[key] = (rs.uniform(0.8, 1.2, size=shape) if randomize_affine
                        else np.ones(shape)).astype(np.float32)
        elif key.endswith(".loc_head.conv1d.bias") and randomize_affine:
            # loc = exp(scale * conv): a bias in [0.5, 3.5] spreads predicted half-lengths over
            # ~2..60 frames so the pooling windows of the fixtures are not all 1-2 frames wide
            lo, hi = cfg.get("loc_bias", (0.5, 3.5))
            out[key] = rs.uniform(lo, hi, size=shape).astype(np.float32)
        elif key.endswith("conv1d.bias") or key.endswith("conv3d.bias"):
            out[key] = (rs.uniform(-0.1, 0.1, size=shape) if randomize_affine
                        else np.zeros(shape)).astype(np.float32)
        elif key.endswith(".weight"):  # GroupNorm / BatchNorm gamma
            out[key] = (rs.uniform(0.5, 1.5, size=shape) if randomize_affine
                        else np.ones(shape)).astype(np.float32)
        elif key.endswith(".bias"):  # GroupNorm / BatchNorm beta
            out[key] = (rs.uniform(-0.2, 0.2, size=shape) if randomize_affine
                        else np.zeros(shape)).astype(np.float32)
        else:
            raise KeyError(key)
    return out


def make_clip(seed, batch, frames=256, crop=96, channels=3):
    """uint8 U[0,255] frames -> x/255*2-1, like AFSD/common/thumos_dataset.py:263."""
    rs = np.random.RandomState(seed)
    u8 = rs.randint(0, 256, size=(batch, channels, frames, crop, crop)).astype(np.uint8)
    return (u8.astype(np.float32) / 255.0) * 2.0 - 1.0


def make_targets(seed, batch, num_classes=15, clip_length=256):
    """Per clip 1..3 GT rows [start, end, label], start<end in [0,1], min length 8 frames."""
    rs = np.random.RandomState(seed)
    out = []
    for _ in range(batch):
        n = rs.randint(1, 4)
        rows = []
        for _ in range(n):
            length = rs.uniform(8.0 / clip_length, 0.6)
            start = rs.uniform(0.0, 1.0 - length)
            rows.append([start, start + length, float(rs.randint(1, num_classes + 1))])
        out.append(np.asarray(rows, np.float32))
    return out


def make_scores_anet(targets, clip_length=768):
    """(b,3,T) [action, start, end] masks in the layout of AFSD/common/anet_dataset.py:250-255 (rows 1 and 2 are
    the ones the training step reads, anet/train.py:136-144); bands as in make_scores."""
    se = make_scores(targets, clip_length)
    act = np.zeros((len(targets), 1, clip_length), np.float32)
    for i, t in enumerate(targets):
        for s, e, _ in t:
            act[i, 0, int(float(s) * clip_length):int(np.ceil(float(e) * clip_length))] = 1.0
    return np.concatenate([act, se], 1)


def make_scores(targets, clip_length=256):
    """start/end boundary masks (b,2,T) as AFSD/common/thumos_dataset.py:110-120 builds them:
    a band of width d = max(len/10, 2) frames centred on each GT boundary is set to 1."""
    b = len(targets)
    sc = np.zeros((b, 2, clip_length), np.float32)
    for i, t in enumerate(targets):
        for s, e, _ in t:
            s_f, e_f = float(s) * clip_length, float(e) * clip_length
            d = max((e_f - s_f) / 10.0, 2.0)
            for ch, ctr in ((0, s_f), (1, e_f)):
                lo = int(np.clip(int(round(ctr - d / 2.0)), 0, clip_length - 1))
                hi = int(np.clip(int(round(ctr + d / 2.0)), 0, clip_length - 1)) + 1
                sc[i, ch, lo:hi] = 1.0
    return sc
