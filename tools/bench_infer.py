"""Config 5: THUMOS14 inference path on 1 MI355X -- proposals/sec = 126 * #clips / time of
(model forward + decode + filter + Soft-NMS).  Synthetic videos; because random weights give
near-uniform scores, the post-processing is ALSO timed alone on synthetic head outputs with
realistic candidate overlap: clustered segments, Beta(0.5,2) scores."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from opental_amd.thumos14 import test as T
from opental_amd.thumos14.BDNet import BDNet


def main():
    nvid = int(sys.argv[1]) if len(sys.argv) > 1 else 16
    from opental_amd.common import ops
    ops.CONV_PRECISION = 1 if (len(sys.argv) > 2 and sys.argv[2] == "bf16") else 0      # default: the exact-fp32 parity path
    dev = torch.device("cuda", 0)
    rs = np.random.RandomState(0)
    torch.manual_seed(0)
    net = BDNet(training=False, use_edl=True)
    net.backbone._model.apply(BDNet.weight_init)
    net = net.to(dev).eval()
    frames = rs.randint(600, 4001, size=nvid)
    g = torch.Generator(device=dev).manual_seed(0)
    videos = [torch.randint(0, 256, (3, int(f), 96, 96), device=dev, generator=g, dtype=torch.uint8) for f in frames]
    nclips = sum(len(T.get_offsets(int(f), 256, 128)) for f in frames)
    T.detect_batch(net, videos[:2], 10.0, batch_clips=32)          # warm-up
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    rows, counts, index, dec = T.detect_batch(net, videos, 10.0, batch_clips=32)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    res = {"conv_operands": "bf16 (fp32 accumulate)" if ops.CONV_PRECISION else "f32 (parity path)", "videos": nvid, "clips": nclips, "end_to_end_s": round(dt, 4),
           "proposals_per_s": round(126 * nclips / dt, 1), "kept": int(counts.sum())}
    # post-processing alone on synthetic head outputs with realistic overlap
    V, C, A, K = 213, 24, 126, 15
    n = V * C
    ctr = torch.rand(n, 12, device=dev, generator=g) * 300
    pick = torch.randint(0, 12, (n, A), device=dev, generator=g)
    c = torch.gather(ctr, 1, pick) + torch.randn(n, A, device=dev, generator=g) * 3
    w = torch.randn(n, A, device=dev, generator=g).abs() * 4 + 6
    sd = dict(seg=torch.stack([c - w / 2, c + w / 2], -1).contiguous(),
              score=torch.distributions.Beta(0.5, 2.0).sample((n, K, A)).to(dev).contiguous(),
              unct=torch.rand(n, A, device=dev, generator=g), actn=torch.rand(n, A, device=dev, generator=g) * 0.6 + 0.4)
    sd["flag"] = ((sd["score"] > 0.01) & (sd["actn"][:, None, :] > 0.5)).to(torch.uint8)
    cs = list(range(0, n + 1, C))
    T.softnms_classes(sd, cs)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); r2, c2, _ = T.softnms_classes(sd, cs); e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1)
    cand = int(sd["flag"].sum())
    res.update({"nms_problems": V * K, "nms_candidates": cand, "nms_ms": round(ms, 3),
                "nms_candidates_per_s": round(cand / ms * 1e3, 1), "nms_kept": int(c2.sum())})
    # CPU oracle on a bounded sample of the same problems (C restatement of softnms_v2)
    from oracle import afsd_oracle as O
    flag = sd["flag"].cpu().numpy().astype(bool); seg = sd["seg"].cpu().numpy(); sc = sd["score"].cpu().numpy()
    t0 = time.perf_counter(); ncpu = 0; kept_cpu = 0
    for v in range(8):
        for k in range(K):
            rows_ = [np.concatenate([seg[ci][flag[ci, k]], sc[ci, k][flag[ci, k], None]], -1) for ci in range(cs[v], cs[v + 1])]
            cnd = torch.from_numpy(np.concatenate(rows_, 0))
            ncpu += len(cnd); kept_cpu += O.softnms_v2_c(cnd)[1]
    dtc = time.perf_counter() - t0
    res.update({"cpu_nms_candidates_per_s": round(ncpu / dtc, 1), "cpu_sample": f"{8 * K} (video,class) problems, C oracle, 1 core"})
    print(json.dumps(res))


if __name__ == "__main__":
    main()
