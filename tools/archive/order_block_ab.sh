run() { python bench.py --steps 10 --warmup 3 --no-cpu-baseline --skip-latency 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$1', '%.0f f/s' % d['value'], 'reference_order %.2f ms' % d['reference_order']['ms_per_step'])"; }
run b256; PWPP_LIB_PATH=$PWD/ab/ord512.so run b512; run b256; PWPP_LIB_PATH=$PWD/ab/ord512.so run b512
PWPP_LIB_PATH=$PWD/ab/ord512.so timeout 300 python -m pytest tests -m gpu -x -q -k "reference_output_order or fuzz" 2>&1 | tail -2
