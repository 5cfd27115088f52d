"""sha256 over the sources that decide what the training step's kernels do and which of them run: every file under
opental_amd/csrc plus the Python that issues the launches.  profiles/*_pmc_step_traffic.json carries the stamp of the tree it
was measured on; bench.py reports `roofline.traffic` from that file only while the stamp still matches.

    python tools/source_stamp.py        -> prints the stamp of this tree
"""
import glob
import hashlib
import os

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LAUNCH_PATH = ("opental_amd/common/ops.py", "opental_amd/common/layers.py", "opental_amd/common/i3d_backbone.py",
               "opental_amd/thumos14/BDNet.py", "opental_amd/thumos14/train.py", "opental_amd/thumos14/multisegment_loss.py",
               "opental_amd/prop_pooling/boundary_pooling_op.py", "opental_amd/thumos14/pyramid_fused.py")


def source_stamp():
    h = hashlib.sha256()
    files = sorted(glob.glob(os.path.join(REPO, "opental_amd", "csrc", "*"))) + [os.path.join(REPO, p) for p in LAUNCH_PATH]
    for f in files:
        if os.path.isfile(f) and not f.endswith((".pyc", ".o", ".so")):
            h.update(os.path.relpath(f, REPO).encode())
            with open(f, "rb") as fh:
                h.update(fh.read())
    return h.hexdigest()[:16]


if __name__ == "__main__":
    print(source_stamp())
