#!/bin/bash
O=gpurun_out/r6r; mkdir -p $O; rm -f $O/*
REPS=2000 N=1 python tools/r6/determinism.py 2>&1 | grep -v amdgpu.ids | tail -3 | cut -c1-300
run() { env $1 python bench.py --modes "" --no-cpu-baseline > $O/bench_$2.$3.json 2>> $O/bench.err; }
for rep in 1 2 3 4 5 6; do run "X=1" new $rep; done
for rep in 1 2 3; do run "TRICOLO_WGRAD_REDUCE_OVERLAP=0" noovl $rep; done
python - <<'P'
import glob, json, collections
res = collections.defaultdict(list)
for f in sorted(glob.glob('gpurun_out/r6r/bench_*.json')):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        r = d['roofline']
        res[f.split('/')[-1].split('.')[0]].append((d['ms_per_step'], d['config']['final_loss'], r['kernel'][:24], r['frac']))
    except Exception as ex:
        print(f, 'ERR', ex)
for k, v in sorted(res.items()):
    for x in v: print(k, x)
P
tail -2 $O/bench.err
python -m pytest tests -m gpu -x -q > $O/tests.txt 2>&1; grep -E "passed|failed|error" $O/tests.txt | tail -3
