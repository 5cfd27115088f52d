import sys; sys.path.insert(0,'/root/repo')
import torch, bench
from opental_amd.common import ops
ops.CONV_PRECISION=1
dev=torch.device('cuda',0)
tr=bench.build_trainer(dev)
ps=[h.scale for h in tr.net.coarse_pyramid_detection.loc_heads]
print([p.data_ptr()-ps[0].data_ptr() for p in ps], [tuple(p.shape) for p in ps], [p.is_contiguous() for p in ps])
print([p.untyped_storage().data_ptr() == ps[0].untyped_storage().data_ptr() for p in ps])
