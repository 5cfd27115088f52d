"""oracle/pin_fusion.py -- TEST INFRASTRUCTURE, runs in the BUILD container only (needs /root/reference).

Pins the two-stream fusion of the inference path (AFSD/thumos14/test.py:79-140: parse_output(fusion=True) +
decode_predictions + filtering) against the imported reference.  The two "networks'" outputs are the two samples of the
reference-generated b = 2 fixture (tests/golden/thumos_b2.npz: out_* of sample 0 = rgb, of sample 1 = flow), so no second
model is needed.  The reference breaks on this configuration (os_head: it adds a squeezed and an unsqueezed actionness,
(126,) + (126,1) -> (126,126)); the flow actionness is therefore handed over already squeezed -- the elementwise average
the code evidently means.  Writes tests/golden/decode_fusion.npz.

    python -m oracle.pin_fusion
"""
import os
import sys

sys.dont_write_bytecode = True
os.environ["PYTHONDONTWRITEBYTECODE"] = "1"
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(REPO, "tests", "golden")

import numpy as np
import torch

from oracle import afsd_oracle as O
from oracle import arch
from oracle.pin_against_reference import REF, import_reference, maxdiff


def main():
    _, _, _, _, ref_test = import_reference()
    fx = np.load(os.path.join(GOLD, "thumos_b2.npz"))
    keys = ("loc", "conf", "prop_loc", "prop_conf", "center", "act", "prop_act", "unct", "prop_unct")
    priors = torch.tensor([[(c + 0.5) / t] for t in arch.level_lengths() for c in range(t)], dtype=torch.float32)
    one = lambda i: dict({k: torch.from_numpy(fx["out_" + k][i:i + 1]) for k in keys}, priors=priors)
    rgb, flow = one(0), one(1)
    flow_ref = dict(flow)
    flow_ref["act"], flow_ref["prop_act"] = flow["act"].squeeze(-1), flow["prop_act"].squeeze(-1)   # see the module docstring
    out_layer = ref_test.DirichletLayer(evidence="exp", dim=-1)
    res = {}
    with torch.no_grad():
        loc, conf, ploc, pconf, center, pri, unct, punct, act, pact = ref_test.parse_output(
            rgb, flow_ref, fusion=True, use_edl=True, os_head=True)
        for idx, (offset, fps) in enumerate(((0, 10.0), (384, 25.0))):
            seg_r, score_r, unct_r, act_r = ref_test.decode_predictions(
                loc, ploc, pri, conf, pconf, unct, punct, act, pact, center, offset, fps, 256, 15,
                score_func=out_layer, use_edl=True, os_head=True)
            seg_o, score_o, unct_o, act_o = O.decode_predictions(O.fuse_outputs(rgb, flow), 0, offset, fps)
            d = max(maxdiff(seg_r, seg_o), maxdiff(score_r, score_o), maxdiff(unct_r, unct_o), maxdiff(act_r, act_o))
            assert d < 1e-6, d
            res[f"seg_{idx}"], res[f"score_{idx}"] = seg_r.numpy().copy(), score_r.numpy().copy()
            res[f"unct_{idx}"], res[f"act_{idx}"] = unct_r.numpy().copy(), act_r.numpy().copy()
            res[f"offset_fps_{idx}"] = np.array([offset, fps], np.float64)
            kept = 0
            for cl in (0, 7, 14):
                fr = ref_test.filtering(seg_r, score_r[cl], unct_r, act_r, 0.01, use_edl=True, os_head=True)
                fo = O.filtering(seg_o, score_o[cl], unct_o, act_o, 0.01)
                assert (fr is None) == (fo is None)
                if fr is not None:
                    assert fr.shape == fo.shape and maxdiff(fr, fo) < 1e-6
                    res[f"filtered_{idx}_{cl}"] = fr.numpy().copy()
                    kept += fr.shape[0]
            print(f"fusion decode {idx}: max diff {d:.2e}, {kept} rows pass the filter in classes 0 / 7 / 14")
        # the uncertainty really is the average of the networks' own maps, not the uncertainty of the averaged logits
        u_avg_logits = O.dirichlet_uncertainty((rgb["conf"][0] + flow["conf"][0]) / 2.0) if hasattr(O, "dirichlet_uncertainty") else None
    np.savez_compressed(os.path.join(GOLD, "decode_fusion.npz"), **res)
    with open(os.path.join(GOLD, "PIN_REPORT.txt"), "a") as f:
        f.write("two-stream fusion: parse_output(fusion=True) + decode_predictions + filtering of the reference on the two samples of "
                "thumos_b2.npz (flow actionness pre-squeezed, see oracle/pin_fusion.py) == O.fuse_outputs + O.decode_predictions -> "
                "tests/golden/decode_fusion.npz\n")
    leftovers = [os.path.join(d_, n) for d_, _, fs in os.walk(REF) for n in fs if n.endswith(".pyc")]
    assert not leftovers, leftovers


if __name__ == "__main__":
    main()
