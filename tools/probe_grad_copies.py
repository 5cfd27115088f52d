import sys; sys.path.insert(0,'/root/repo')
import torch, bench
from opental_amd.common import ops
ops.CONV_PRECISION=1
dev=torch.device('cuda',0)
tr=bench.build_trainer(dev)
clips,targets,scores=bench.synth_batch(8,1000,dev)
for _ in range(2): tr.step(clips,targets,scores)
names={id(p):n for n,p in tr.net.named_parameters()}
orig=tr._flush_bucket
def spy(b):
    a=tr.arena
    if not tr._flushed[b]:
        for i in a.bucket_members[b]:
            if tr._in_arena[i]: continue
            g=a.params[i].grad
            if g is not None and g.data_ptr()!=a.grad_views[i].data_ptr():
                print("copy", b, names[id(a.params[i])], tuple(a.params[i].shape))
    return orig(b)
tr._flush_bucket=spy
tr.step(clips,targets,scores)
torch.cuda.synchronize()
