"""The CPU restatement regenerates the committed golden vectors (which were produced by the
imported reference in the build container, oracle/pin_against_reference.py) from seeds alone."""
import os

import numpy as np
import pytest
import torch

from oracle import afsd_oracle as O
from oracle import arch


@pytest.fixture(scope="module")
def b1(golden_dir):
    fx = np.load(os.path.join(golden_dir, "thumos_b1.npz"))
    P = O.to_torch(arch.make_params(int(fx["param_seed"])))
    x = torch.from_numpy(arch.make_clip(int(fx["clip_seed"]), 1))
    keep = {}
    torch.set_num_threads(os.cpu_count())
    with torch.no_grad():
        out = O.bdnet_forward(P, x, keep=keep)
    return fx, out, keep


def test_spec_counts():
    spec = arch.param_spec()
    assert len(spec) == 446
    n = sum(int(np.prod(s)) for k, s in spec if not k.endswith("num_batches_tracked"))
    assert n == 44750436 - 57  # 57 num_batches_tracked scalars are not float parameters


def test_forward_matches_reference_vectors(b1):
    fx, out, keep = b1
    for k in ("loc", "conf", "prop_loc", "prop_conf", "center", "act", "prop_act", "unct", "prop_unct"):
        np.testing.assert_allclose(out[k].numpy(), fx["out_" + k], rtol=0, atol=1e-6, err_msg=k)
    for i in range(6):
        assert np.array_equal(keep["segments"][i].numpy(), fx[f"segments_{i}"])
        assert np.array_equal(keep["frame_segments"][i].numpy(), fx[f"frame_segments_{i}"])


def test_losses_match_reference_vectors(b1):
    fx, out, _ = b1
    targets = [torch.from_numpy(fx["target_0"])]
    for mode, epoch in (("edl", 0), ("edl", 12), ("focal", 0)):
        st = O.EvidenceState()
        st.epoch = epoch
        got = O.multisegment_loss(out, targets, cls_loss_type=mode, state=st if mode == "edl" else None)
        np.testing.assert_allclose([float(v) for v in got], fx[f"loss_{mode}{epoch}"], rtol=1e-6, atol=1e-6)
        if epoch >= 10:
            np.testing.assert_allclose(st.weight_accum.numpy(), fx["loss_edl12_weight_accum"], atol=1e-7)
    cost, _ = O.train_cost(out, targets, torch.from_numpy(fx["scores"]), state=O.EvidenceState())
    assert abs(float(cost) - float(fx["cost_edl0"])) < 1e-4


def test_softnms_vectors(golden_dir):
    fx = np.load(os.path.join(golden_dir, "softnms.npz"))
    for n in (0, 1, 2, 50, 400, 2000):
        seg = torch.from_numpy(fx[f"in_{n}"])
        for impl in (O.softnms_v2, O.softnms_v2_c):
            rows, cnt, mask = impl(seg.clone())
            assert np.array_equal(mask.numpy(), fx[f"mask_{n}"]), (n, impl.__name__)
            if cnt:
                np.testing.assert_allclose(rows.numpy(), fx[f"rows_{n}"], atol=1e-6)
    rows, cnt, mask = O.softnms_v2_c(torch.from_numpy(fx["in_400"]), top_k=20)
    assert cnt == 20 and np.array_equal(mask.numpy(), fx["mask_400_top20"])


def test_softnms_quirks():
    # a single candidate is never kept; output is in index order (segment_utils.py:136,157-159)
    one = torch.tensor([[0., 1., 0.9, 0.1, 0.9]])
    assert O.softnms_v2(one)[1] == 0 and O.softnms_v2_c(one)[1] == 0
    three = torch.tensor([[0., 1., 0.2, 0, 0], [5., 6., 0.9, 0, 0], [10., 11., 0.5, 0, 0]])
    rows, cnt, mask = O.softnms_v2(three)
    assert cnt == 2 and mask.tolist() == [False, True, True] and rows[0, 2] == 0.9 and rows[1, 2] == 0.5


def test_ssl_triplet_branch_matches_reference_vectors(golden_dir):
    """a14: O.ssl_triplets / O.triplet_cost regenerate what the reference's BDNet.forward(ssl=True) and its three
    TripletMarginLoss terms gave in the build container (fixture entries ssl{set}_*, oracle/pin_against_reference.py)."""
    fx = np.load(os.path.join(golden_dir, "thumos_b1.npz"))
    P = O.to_torch(arch.make_params(int(fx["param_seed"])))
    x = torch.from_numpy(arch.make_clip(int(fx["ssl_clip_seed"]), 1))
    torch.set_num_threads(os.cpu_count())
    for si in range(fx["ssl_proposals"].shape[0]):
        props = [torch.from_numpy(fx["ssl_proposals"][si])]
        with torch.no_grad():
            a, p, n = O.ssl_triplets(P, x, props)
            cost = float(O.triplet_cost(a, p, n, 1.0))
        for nm, got in (("anchor", a), ("positive", p), ("negative", n)):
            for i in range(3):
                np.testing.assert_allclose(got[i].numpy(), fx[f"ssl{si}_{nm}_{i}"], rtol=0, atol=1e-6)
        assert abs(cost - float(fx[f"ssl{si}_cost"])) < 1e-5
        assert abs(cost - float(fx[f"ssl{si}_terms"].sum())) < 1e-5


def test_two_stream_fusion_matches_reference_vectors(golden_dir):
    """O.fuse_outputs + O.decode_predictions regenerate the reference's parse_output(fusion=True) + decode_predictions
    (tests/golden/decode_fusion.npz, oracle/pin_fusion.py) from the two samples of the b = 2 fixture."""
    fx, want = np.load(os.path.join(golden_dir, "thumos_b2.npz")), np.load(os.path.join(golden_dir, "decode_fusion.npz"))
    keys = ("loc", "conf", "prop_loc", "prop_conf", "center", "act", "prop_act", "unct", "prop_unct")
    priors = torch.tensor([[(c + 0.5) / t] for t in arch.level_lengths() for c in range(t)], dtype=torch.float32)
    one = lambda i: dict({k: torch.from_numpy(fx["out_" + k][i:i + 1]) for k in keys}, priors=priors)
    fused = O.fuse_outputs(one(0), one(1))
    for idx in (0, 1):
        offset, fps = want[f"offset_fps_{idx}"]
        seg, score, unct, act = O.decode_predictions(fused, 0, float(offset), float(fps))
        for got, k in ((seg, "seg"), (score, "score"), (unct, "unct"), (act, "act")):
            np.testing.assert_allclose(got.numpy(), want[f"{k}_{idx}"], rtol=0, atol=1e-6)
