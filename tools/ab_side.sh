# A/B: weight gradients on the side stream, eager launches: chunk size of the deferred issue
run() { echo "== $* $EXTRA"; env "$@" python bench.py --no-extras --no-cpu-baseline --no-hbm-kernels --no-roofline --steps 20 $EXTRA 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['config'].get('launch'), d['config'].get('launch_probe'))"; }
python -m pytest tests/test_determinism_gpu.py tests/test_model_gpu.py -m gpu -q -x 2>&1 | tail -3
for c in 1 4 6 10 16; do
EXTRA="--graph off" run OTAL_WGRAD_CHUNK=$c
done
EXTRA="--batch 4 --graph off" run OTAL_WGRAD_CHUNK=6
EXTRA="--batch 4 --graph off" run OTAL_WGRAD_STREAM=0
