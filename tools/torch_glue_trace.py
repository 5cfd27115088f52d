"""Attribute the aten ops of one training step to the Python lines that issue them (TorchDispatchMode + traceback).
usage: python tools/torch_glue_trace.py [batch]   -- ops issued from the autograd engine show the backward node's frame"""
import os, sys, collections, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.utils._python_dispatch import TorchDispatchMode
import bench
from opental_amd.common import ops

SKIP = ("aten.view", "aten.detach", "aten.slice", "aten.select", "aten._unsafe_view", "aten.reshape", "aten.t.", "aten.alias",
        "aten.expand", "aten.permute", "aten.transpose", "aten.unsqueeze", "aten.squeeze", "aten.as_strided", "aten.empty",
        "aten.split", "aten.unbind", "aten.is_", "aten.lift", "aten.sym_", "aten.stride", "aten.size", "aten._local_scalar", "aten.narrow")


class Log(TorchDispatchMode):
    def __init__(self):
        super().__init__()
        self.by = collections.Counter()

    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = str(func)
        if not name.startswith(SKIP):
            src = "<autograd engine / C++>"
            for fr in reversed(traceback.extract_stack()):
                if ("/opental_amd/" in fr.filename or fr.filename.endswith("bench.py")) and "tools/" not in fr.filename:
                    src = f"{fr.filename.split('/root/repo/')[-1].split('opental_amd/')[-1]}:{fr.lineno} {fr.name}"
                    break
            self.by[(name, src)] += 1
        return func(*args, **(kwargs or {}))


def main():
    batch = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    ops.CONV_PRECISION = 1
    dev = torch.device("cuda", 0)
    tr = bench.build_trainer(dev)
    clips, targets, scores = bench.synth_batch(batch, 1000, dev)
    for _ in range(3):
        tr.step(clips, targets, scores)
    torch.cuda.synchronize()
    log = Log()
    with log:
        tr.step(clips, targets, scores)
    torch.cuda.synchronize()
    print("aten ops that may launch:", sum(log.by.values()))
    for (name, src), n in log.by.most_common(90):
        print(f"{n:4d}  {name:34s} {src}")


if __name__ == "__main__":
    main()
