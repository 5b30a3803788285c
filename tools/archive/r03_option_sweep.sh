for v in "X=1" "PWPP_BIN_BLOCK=512" "PWPP_FIT_STREAMS=1" "PWPP_FIT_STREAMS=3" "PWPP_OVERLAP_RANGES=3" "PWPP_FIT_CONCURRENT=1" "PWPP_HI_SPLIT=0.5" "PWPP_HI_SPLIT=0.7" "X=1"; do
  env $v python bench.py --steps 40 --warmup 5 --no-cpu-baseline --skip-latency --skip-extras --profile-steps 1 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%-24s'%'$v', round(d['value']), round(d['ms_per_step'],3))"
done
