run() { echo -n "$1: "; env $1 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --skip-extras --skip-latency 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('%.0f f/s  %.3f ms/step' % (d['value'], d['ms_per_step']))"; }
run "X=0"
run "PWPP_FIT_CONCURRENT=1"
run "PWPP_OVERLAP=0 PWPP_FIT_CONCURRENT=1"
run "PWPP_OVERLAP=0"
run "PWPP_FIT_STREAMS=4"
