for plan in "$@"; do
  echo "== $plan"
  PWPP_FIT_PLAN=$plan python bench.py --steps 5 --warmup 1 --no-cpu-baseline --skip-latency 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d[\"value\"]), round(d[\"ms_per_step\"],3), {k:round(v,3) for k,v in d[\"kernel_ms\"].items() if \"fit\" in k})"
done
