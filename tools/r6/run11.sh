#!/bin/bash
O=gpurun_out/r6l; mkdir -p $O; rm -f $O/*
python -m pytest tests -m gpu -x -q -k "adam or halo or training_steps or graph or dp" > $O/tests.txt 2>&1; tail -5 $O/tests.txt
run() { env $1 python bench.py --modes "" --no-cpu-baseline > $O/bench_$2.$3.json 2>> $O/bench.err; }
for rep in 1 2 3; do
  run "TRICOLO_DS_ON_TEXT=0" base $rep
  run "TRICOLO_DS_ON_TEXT=1" lend $rep
done
python - <<'P'
import glob, json, collections
res = collections.defaultdict(list)
for f in sorted(glob.glob('gpurun_out/r6l/bench_*.json')):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        r = d['roofline']
        res[f.split('/')[-1].split('.')[0]].append((d['ms_per_step'], r['kernel'][:24], r['frac'], r['avg_launch_ms']))
    except Exception as ex:
        print(f, 'ERR', ex)
for k, v in sorted(res.items()):
    for x in v: print(k, x)
P
tail -3 $O/bench.err
python tools/step_timeline.py 2>/dev/null | grep -E "step.start|fwd.end|loss|bwd.start|bwd.end|adam|step.end"
