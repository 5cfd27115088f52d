# A/B: the Inception modules' small branches on the branch lane (eager launches)
run() { echo "== $* $EXTRA"; env "$@" python bench.py --no-extras --no-cpu-baseline --no-hbm-kernels --no-roofline --steps 30 $EXTRA 2>/tmp/ab.err | tail -1 > /tmp/ab.out; python -c "import json,sys; d=json.loads(open('/tmp/ab.out').read()); print(d['value'], d['ms_per_step'], d['config'].get('launch'), d['config'].get('launch_probe'))" 2>/dev/null || grep -v "^  File\|^    \|^frame" /tmp/ab.err | tail -12; }
python -m pytest tests/test_determinism_gpu.py tests/test_model_gpu.py tests/test_train_gpu.py -m gpu -q -x 2>&1 | tail -3
EXTRA="--graph off" run OTAL_BRANCH_LANE=1
EXTRA="--graph off" run OTAL_BRANCH_LANE=0
EXTRA="--graph off" run OTAL_BRANCH_LANE=1
EXTRA="--graph off" run OTAL_BRANCH_LANE=0
EXTRA="--graph lanes" run OTAL_BRANCH_LANE=1
