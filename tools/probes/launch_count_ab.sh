#!/bin/bash
# Does the NUMBER of side-tower launches move the trimodal step?  Default against TRICOLO_NO_MASK_PYRAMID=1 (7 more short launches on the voxel tower, same work),
# alternating, N pairs; prints every run and the two medians.
N=${1:-8}
a=(); b=()
for i in $(seq $N); do
  x=$(python bench.py --modes "" --no-cpu-baseline 2>/dev/null | python -c "import json,sys; print(json.loads(sys.stdin.read())['ms_per_step'])")
  y=$(TRICOLO_NO_MASK_PYRAMID=1 python bench.py --modes "" --no-cpu-baseline 2>/dev/null | python -c "import json,sys; print(json.loads(sys.stdin.read())['ms_per_step'])")
  echo "pair $i: default $x  +7 launches $y"
  a+=($x); b+=($y)
done
python - "${a[@]}" -- "${b[@]}" <<'PY'
import sys, statistics
args = sys.argv[1:]
k = args.index("--")
a, b = list(map(float, args[:k])), list(map(float, args[k + 1:]))
print("median default", statistics.median(a), " median +7 launches", statistics.median(b), " mean diff", round(statistics.mean(b) - statistics.mean(a), 4), "ms")
PY
