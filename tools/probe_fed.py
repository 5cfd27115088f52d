"""Where the time of a pipeline-fed training loop goes (host side): python tools/probe_fed.py [--batch 8]
Prints ms per step and the host's share per call (collect / submit / step) for the lane-graph step fed by the ClipStager,
next to the same step fed from resident buffers."""
import os
import sys
import time

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import bench  # noqa: E402
from opental_amd.common import ops, thumos_dataset as D  # noqa: E402


def main():
    batch = int(sys.argv[sys.argv.index("--batch") + 1]) if "--batch" in sys.argv else 8
    dev = torch.device("cuda", 0)
    ops.CONV_PRECISION = 1
    rs = np.random.RandomState(1)
    vids = [torch.from_numpy(rs.randint(0, 256, (1200, 112, 112, 3)).astype(np.uint8)).pin_memory() for _ in range(4)]
    # OTAL_PROBE_DUMMY_STREAMS=n: n streams created (and used once) first -- HIP hands its few hardware queues to streams in
    # creation order, so this moves the stager's copy stream onto another hardware queue
    dummies = [torch.cuda.Stream(device=dev) for _ in range(int(os.environ.get("OTAL_PROBE_DUMMY_STREAMS", "0")))]
    for d_ in dummies:
        with torch.cuda.stream(d_):
            torch.zeros(8, device=dev)
    side = ops.side_wgrads(dev).side if os.environ.get('OTAL_PROBE_COPY_ON_SIDE', '1') != '0' else None
    st = D.ClipStager(batch, 256, 112, 112, 96, device=dev, max_targets=8, score_rows=2, copy_stream=side)

    def samples(k):
        r = np.random.RandomState(k)
        out = []
        for _ in range(batch):
            n = int(r.randint(1, 7))
            a = np.sort(r.uniform(0.0, 1.0, (n, 2)), 1)
            a[:, 1] = np.maximum(a[:, 1], a[:, 0] + 8.0 / 256)
            tg = np.concatenate([a, r.randint(1, 16, (n, 1))], 1).astype(np.float32)
            out.append({"video": vids[int(r.randint(4))], "offset": int(r.randint(0, 900)), "frame_map": None,
                        "crop": (int(r.randint(17)), int(r.randint(17)), bool(r.randint(2))), "target": tg,
                        "scores": (r.uniform(size=(2, 256)) < 0.05).astype(np.float32)})
        return out
    pre = [samples(300 + k) for k in range(64)]
    tr = bench.build_trainer(dev)
    tr.launch = 'lanes'
    st.submit(pre[0])
    for k in range(5):
        static = tr.static_inputs()
        clips, _ = st.collect(out=None if static is None else static[0])
        rec = st.labels()
        st.submit(pre[k + 1])
        tr.step(clips, rec.targets, rec.scores)
        st.release()
    torch.cuda.synchronize()

    def loop(n, do_submit=True, do_collect=True):
        t_c = t_s = t_p = 0.0
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k in range(n):
            a = time.perf_counter()
            if do_collect:
                clips, _ = st.collect(out=tr.static_inputs()[0])
                rec = st.labels()
            else:
                clips, rec = tr.static_inputs()[0], st.labels_dev[0]
            b = time.perf_counter()
            if do_submit:
                st.submit(pre[(k + 7) % 64])
            c = time.perf_counter()
            tr.step(clips, rec.targets, rec.scores)
            if do_collect:
                st.release()
            d = time.perf_counter()
            t_c += b - a; t_s += c - b; t_p += d - c
        host = time.perf_counter() - t0
        torch.cuda.synchronize()
        tot = time.perf_counter() - t0
        return tot / n * 1e3, host / n * 1e3, t_c / n * 1e3, t_s / n * 1e3, t_p / n * 1e3

    for name, kw in (("stager-fed", {}), ("resident (no collect / submit)", dict(do_submit=False, do_collect=False)),
                     ("stager-fed again", {})):
        if not kw:
            st.submit(pre[6])
        r = loop(30, **kw)
        if not kw:
            st.collect()
        print(f"{name}: {r[0]:.3f} ms/step = {batch / r[0] * 1e3:.0f} clips/s; host issue {r[1]:.3f} ms "
              f"(collect {r[2]:.3f}, submit {r[3]:.3f}, step {r[4]:.3f})", flush=True)


if __name__ == "__main__":
    main()
