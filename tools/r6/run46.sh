#!/bin/bash
O=gpurun_out/r6aq; mkdir -p $O
for rep in 1 2 3 4 5 6 7 8; do
timeout 200 python tools/step_timeline.py > $O/tl.$rep.txt 2>/dev/null
grep -E "step.end" $O/tl.$rep.txt | head -1
done
for rep in 1 2 3 4 5 6 7 8; do
timeout 200 python bench.py --modes "" --no-cpu-baseline --no-voxel-config5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench', d['ms_per_step'])"
done
