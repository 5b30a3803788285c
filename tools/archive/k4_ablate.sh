#!/bin/bash
# timing ablations of k_patch_fit (results are wrong by construction when flags are set)
for f in 0 1 2 3; do
  echo "== PWPP_DEBUG_FLAGS=$f"
  PWPP_DEBUG_FLAGS=$f python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['kernel_ms'])"
done
