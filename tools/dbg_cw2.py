import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from opental_amd.common import ops
ops.CONV_PRECISION = 1
B, C, T, H, W = 1, 64, 2, 12, 12
x = torch.zeros(B, C, T, H, W, device="cuda")
wv = torch.arange(W, device="cuda").float()
for c in range(C):
    x[0, c] = (wv + 16 * c)[None, None, :]           # value encodes (channel, w)
for tap in ((1, 1, 1), (1, 1, 0), (1, 1, 2)):
    for ci in (0, 1, 5, 9):
        w = torch.zeros(192, C, 3, 3, 3, device="cuda")
        w[0, ci, tap[0], tap[1], tap[2]] = 1.0
        y = ops.conv_forward(x, w, (3, 3, 3), (1, 1, 1))
        print("tap", tap, "ci", ci, "y[0,0,0,3,:] =", y[0, 0, 0, 3, :].tolist())
