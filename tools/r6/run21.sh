#!/bin/bash
O=gpurun_out/r6s; mkdir -p $O; rm -f $O/*
python -m pytest tests -m gpu -x -q -k "training_steps or f16_mode or graph or reproducible or mvcnn" > $O/tests.txt 2>&1; grep -E "passed|failed|error" $O/tests.txt | tail -3
run() { env $1 python bench.py --modes "" --no-cpu-baseline > $O/bench_$2.$3.json 2>> $O/bench.err; }
for rep in 1 2 3 4; do run "X=1" beside $rep; run "TRICOLO_STEM_BESIDE_WGRAD=0" serial $rep; done
python - <<'P'
import glob, json, collections
res = collections.defaultdict(list)
for f in sorted(glob.glob('gpurun_out/r6s/bench_*.json')):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        res[f.split('/')[-1].split('.')[0]].append((d['ms_per_step'], d['config']['final_loss']))
    except Exception as ex:
        print(f, 'ERR', ex)
for k, v in sorted(res.items()): print(k, v)
P
tail -2 $O/bench.err
python tools/step_timeline.py 2>/dev/null | grep -E "step.start|fwd.end|loss|bwd|adam|step.end"
