import os, sys, faulthandler
faulthandler.enable()
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tools')
import tempfile
from make_synthetic_thumos import make
from opental_amd.thumos14 import train as R
uniform = int(sys.argv[1]); steps = sys.argv[2]
d = tempfile.mkdtemp()
y = make(d + "/data", videos=3, frames=520, size=100, uniform=uniform)
FLAGS = ['--open_set', '--split', '0', '--lw', '1', '--cw', '10', '--piou', '0.5', '--ssl', '0.001', '--batch_size', '2']
tr, hist = R.main([y] + FLAGS + ['--random_init', '--max_steps', steps, '--max_epoch', '1', '--checkpoint_path', d + "/run"] + sys.argv[3:])
print("ok", tr.replayed_steps, hist, flush=True)
