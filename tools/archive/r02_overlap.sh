#!/bin/bash
cd $GRAFT_REPO_ROOT
for r in "$@"; do
  echo "== overlap ranges $r"
  PWPP_OVERLAP_RANGES=$r python bench.py --steps 20 --warmup 3 --no-cpu-baseline --skip-latency --overlap --no-profile-events 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']), round(d['ms_per_step'],3))"
done
