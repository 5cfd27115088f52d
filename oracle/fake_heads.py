"""oracle/fake_heads.py -- TEST INFRASTRUCTURE.  A stand-in detector for driver-level parity tests: head outputs that
are a deterministic function of the WINDOW CONTENT (so the reference's per-window b=1 calls and this package's batched
calls see the same outputs for the same window, in whatever order or batch they arrive), with realistic ranges.
Used by oracle/pin_cross_data.py (imports the reference, build container only) and by tests/test_cross_data.py."""
import numpy as np
import torch

from oracle import afsd_oracle as O
from oracle import arch


def window_seeds(clips):
    """One integer per window: the exact sum of its uint8 pixel values (recovered from the normalised floats), which
    does not depend on summation order or device."""
    u = torch.round((clips.double() + 1.0) * 127.5).to(torch.int64)
    return [int(v) % 2147483629 for v in u.reshape(u.shape[0], -1).sum(1).cpu()]


def heads_from_seed(seed, cfg=arch.THUMOS):
    rs = np.random.RandomState(seed)
    pri = O.priors_all(cfg)
    K, C = pri.shape[0], cfg["num_classes"]
    f = lambda a: torch.from_numpy(a.astype(np.float32))
    return {"loc": f(rs.uniform(2.0, 60.0, (K, 2))), "conf": f(rs.normal(0, 2.0, (K, C))),
            "prop_loc": f(rs.normal(0, 0.3, (K, 2))), "prop_conf": f(rs.normal(0, 2.0, (K, C))),
            "center": f(rs.normal(-1.0, 1.0, (K, 1))), "act": f(rs.normal(0.5, 1.0, (K, 1))),
            "prop_act": f(rs.normal(0.5, 1.0, (K, 1)))}


class FakeNet:
    """net(clips (n,3,T,H,W) in [-1,1]) -> the detector's output dict for n windows, on the clips' device."""

    def __init__(self, cfg=arch.THUMOS):
        self.cfg = cfg
        self.calls = 0

    def eval(self):
        return self

    def __call__(self, clips):
        self.calls += 1
        per = [heads_from_seed(s, self.cfg) for s in window_seeds(clips)]
        out = {k: torch.stack([p[k] for p in per], 0).to(clips.device) for k in per[0]}
        out["priors"] = O.priors_all(self.cfg).to(clips.device)
        C = self.cfg["num_classes"]
        for k, src in (("unct", "conf"), ("prop_unct", "prop_conf")):
            out[k] = C / (torch.exp(torch.clamp(out[src], -10, 10)) + 1).sum(-1)      # DirichletLayer.compute_uncertainty
        return out


def synthetic_video(seed, frames, size=96):
    """uint8 (frames, size, size, 3) -- the on-disk layout of the reference's .npy clips (test.py:59-64)."""
    rs = np.random.RandomState(seed)
    return rs.randint(0, 256, size=(frames, size, size, 3)).astype(np.uint8)


THUMOS_TRAIN = (("video_validation_0000051", 301, 300, 10.0), ("video_validation_0000052", 302, 640, 10.0),
                ("video_validation_0000053", 303, 200, 10.0))                  # name, seed, frames, sample fps
ANET_VIDEOS = (("v_aaa111", 401, 300, 10.0, 31.0, ["Knitting"]), ("v_bbb222", 402, 700, 12.5, 52.0, ["Long jump"]),
               ("v_ccc333", 403, 256, 8.0, 33.5, ["Painting", "Knitting"]), ("v_ddd444", 404, 130, 10.0, 9.0, []))
#              name, seed, frames, fps, duration (s; SHORTER than frames / fps for two of them: clipping), labels
OVERLAPPING = ["Long jump", "Shot put"]
