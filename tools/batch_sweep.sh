# clips/s, ms/step, conv TFLOP/s and roofline fraction of the training step at per-GPU batches 1 .. 32 (run through gpurun)
for b in 1 2 4 8 16 32; do
  r=$(python bench.py --batch $b --no-extras --no-cpu-baseline --no-hbm-kernels --steps 20 --warmup 5 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print(d['value'], d['ms_per_step'], r['achieved'], r['frac'], d['config']['launch'][:30])")
  echo "b=$b $r"
done
