#!/bin/bash
# round 6, GPU call 2: shortcut-branch schedule A/B (TRICOLO_DS_LATE 0 / 1 / 2), module parity tests on the new default
O=gpurun_out/r6b; mkdir -p $O
for rep in 1 2; do
for e in "TRICOLO_DS_LATE=0" "TRICOLO_DS_LATE=1" "TRICOLO_DS_LATE=2"; do
  env $e python bench.py --modes "" --no-cpu-baseline > $O/bench_$e.$rep.json 2>> $O/bench.err
done; done
python - <<'P'
import glob, json
for f in sorted(glob.glob('gpurun_out/r6b/bench_*.json')):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        r = d['roofline']
        print(f, d['ms_per_step'], d['ms_per_step_windows']['min'], r['kernel'][:30], r['frac'], r['avg_launch_ms'])
    except Exception as ex:
        print(f, 'ERR', ex)
P
python tools/step_timeline.py > $O/timeline_late2.txt 2>/dev/null; cat $O/timeline_late2.txt
TRICOLO_DS_LATE=0 python tools/step_timeline.py > $O/timeline_late0.txt 2>/dev/null; cat $O/timeline_late0.txt
python -m pytest tests -m gpu -x -q > $O/tests_all.txt 2>&1; tail -5 $O/tests_all.txt
