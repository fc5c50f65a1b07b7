#!/bin/bash
# round 6, GPU call 6: in-graph kernel trace with the new issue orders
O=gpurun_out/r6f; mkdir -p $O
python bench.py --modes "" --no-cpu-baseline > $O/bench.json 2> $O/bench.err
python - <<'P'
import json
d = json.loads(open('gpurun_out/r6f/bench.json').read().strip().splitlines()[-1])
r = d['roofline']
print(d['ms_per_step'], r['kernel'][:40], r['frac'], r['frac_raw_events'], r['avg_launch_ms'])
for k, v in r['families'].items(): print(' ', k, v)
P
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/trace -- python3 $GRAFT_REPO_ROOT/bench.py --modes "" --no-cpu-baseline --steps 10 --warmup 3 --windows 1 > $GRAFT_REPO_ROOT/$O/trace_bench.json 2> $GRAFT_REPO_ROOT/$O/trace_bench.err
cd $GRAFT_REPO_ROOT
python tools/trace_step.py $O/trace --shortest > $O/trace_step.txt 2>&1
find $O/trace -name "*.csv" -size +1M -delete
tail -3 $O/trace_step.txt
