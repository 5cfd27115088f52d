# A/B: eager two-stream launches vs the lane-graph replay
run() { echo "== $* $EXTRA"; env "$@" python bench.py --no-extras --no-cpu-baseline --no-hbm-kernels --no-roofline --steps 20 $EXTRA 2>/tmp/ab.err | tail -1 > /tmp/ab.out; python -c "import json,sys; d=json.loads(open('/tmp/ab.out').read()); print(d['value'], d['ms_per_step'], d['config'].get('launch'), d['config'].get('launch_probe'))" 2>/dev/null || grep -v "^  File\|^    \|^frame" /tmp/ab.err | tail -12; }
EXTRA="--graph lanes" run OTAL_WGRAD_CHUNK=2
EXTRA="--graph lanes" run OTAL_WGRAD_CHUNK=8
EXTRA="--graph lanes" run OTAL_WGRAD_CHUNK=16
EXTRA="--graph lanes --batch 1" run OTAL_WGRAD_CHUNK=4
EXTRA="--graph on --batch 1" run OTAL_WGRAD_CHUNK=4
EXTRA="--graph lanes --batch 2" run OTAL_WGRAD_CHUNK=4
EXTRA="--graph on --batch 2" run OTAL_WGRAD_CHUNK=4
EXTRA="--graph lanes --batch 16" run OTAL_WGRAD_CHUNK=4
EXTRA="--graph off --batch 16" run OTAL_WGRAD_CHUNK=4
