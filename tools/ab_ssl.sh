# A/B of the ssl-on-every-step leg on ONE box: tools/ab_ssl.sh "OTAL_LIB_PATH=ab/x.so" "-"
for cfg in "$@"; do
  [ "$cfg" = "-" ] && cfg=""
  for rep in 1 2; do
    r=$(env $cfg python bench.py --ssl --no-extras --no-cpu-baseline --no-hbm-kernels --no-roofline --steps 30 --warmup 5 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'], d['config'].get('launch'))")
    echo "[$cfg] $r"
  done
done
