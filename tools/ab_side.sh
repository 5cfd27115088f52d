# A/B: eager vs hybrid (eager backbone + captured heads) vs lane graphs
run() { echo "== $* $EXTRA"; env "$@" python bench.py --no-extras --no-cpu-baseline --no-hbm-kernels --no-roofline --steps 30 $EXTRA 2>/tmp/ab.err | tail -1 > /tmp/ab.out; python -c "import json,sys; d=json.loads(open('/tmp/ab.out').read()); print(d['value'], d['ms_per_step'], d['config'].get('launch')[:40], d['config'].get('launch_probe'))" 2>/dev/null || grep -v "^  File\|^    \|^frame" /tmp/ab.err | tail -12; }
EXTRA="--graph hybrid" run A=1
EXTRA="--graph off" run A=1
EXTRA="--graph lanes" run A=1
EXTRA="--graph hybrid" run A=1
EXTRA="--graph off" run A=1
