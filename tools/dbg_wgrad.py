import sys; sys.path.insert(0,'/root/repo')
import numpy as np, torch, torch.nn.functional as F
from opental_amd.common import ops
from oracle import afsd_oracle as O
rs=np.random.RandomState(0)
for (B,C,T,Co,k,s) in ((1,512,256,512,3,1),(2,512,256,512,3,1),(1,512,256,512,1,1),(1,512,64,512,3,1),(1,512,126,512,3,1)):
    x=torch.from_numpy(rs.randn(B,C,T).astype(np.float32)); w=torch.from_numpy((rs.randn(Co,C,k)/40).astype(np.float32))
    xr=x.clone().requires_grad_(True); wr=w.clone().requires_grad_(True)
    y=O.unit1d(xr,wr,None,s); dy=torch.from_numpy(rs.randn(*y.shape).astype(np.float32)); y.backward(dy)
    for rep in range(2):
        dw=ops.conv_wgrad(x.cuda(),dy.cuda(),w.shape,k,s).cpu()
        dx=ops.conv_dgrad(dy.cuda(),w.cuda(),x.shape,k,s).cpu()
        yy=ops.conv_forward(x.cuda(),w.cuda(),k,s).cpu()
        print((B,C,T,Co,k,s),rep,'fwd',float((yy-y.detach()).abs().max()/y.abs().max()),'dw',float((dw-wr.grad).abs().max()/wr.grad.abs().max()),'dx',float((dx-xr.grad).abs().max()/xr.grad.abs().max()))
