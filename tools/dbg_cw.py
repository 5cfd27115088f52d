"""Debug: bf16 conv fwd/dgrad of one geometry vs torch on bf16-rounded operands; prints where the error sits."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F
from opental_amd.common import ops
ops.CONV_PRECISION = 1
shape = tuple(int(v) for v in sys.argv[1].split(",")) if len(sys.argv) > 1 else (1, 64, 6, 12, 12)
cout = int(sys.argv[2]) if len(sys.argv) > 2 else 192
torch.manual_seed(0)
x = torch.randn(*shape, device="cuda").bfloat16().float()
w = (torch.randn(cout, shape[1], 3, 3, 3, device="cuda") * 0.05).bfloat16().float()
y = ops.conv_forward(x, w, (3, 3, 3), (1, 1, 1))
ref = F.conv3d(x, w, padding=1)
err = (y - ref).abs()
print("fwd max err", float(err.max()), "scale", float(ref.abs().max()))
bad = (err > 1e-3 * ref.abs().max()).nonzero()
print("bad count", len(bad), "of", err.numel())
if len(bad):
    for name, d in zip("bcthw", range(5)):
        print(name, torch.unique(bad[:, d]).tolist()[:40])
dy = torch.randn_like(ref).bfloat16().float()
dx = ops.conv_dgrad(dy, w, x.shape, (3, 3, 3), (1, 1, 1))
refdx = F.conv_transpose3d(dy, w, padding=1)
e2 = (dx - refdx).abs()
print("dgrad max err", float(e2.max()), "scale", float(refdx.abs().max()))
bad = (e2 > 1e-3 * refdx.abs().max()).nonzero()
print("bad count", len(bad))
if len(bad):
    for name, d in zip("bcthw", range(5)):
        print(name, torch.unique(bad[:, d]).tolist()[:40])
