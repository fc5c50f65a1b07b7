#!/bin/bash
# bench.py default workload under the tower issue orders (TRICOLO_TOWER_ORDER), alternating, two passes
for rep in 1 2; do
  for o in itv tvi vti tiv; do
    TRICOLO_TOWER_ORDER=$o python bench.py --modes "" --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$o', d['ms_per_step'], d['value'])"
  done
done
