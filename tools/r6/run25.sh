#!/bin/bash
O=gpurun_out/r6w; mkdir -p $O; rm -f $O/*
timeout 900 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "adam or overflow or wgrad" 2>&1 | tail -4
timeout 900 python -m pytest tests/test_gpu_modules.py -x -q -m gpu -k "guard or reproducible or graph or dp_" 2>&1 | tail -4
run() { env $1 python bench.py --modes "" --no-cpu-baseline --no-voxel-config5 > $O/bench_$2.$3.json 2>> $O/bench.err; }
for rep in 1 2 3; do run "X=1" new $rep; done
python - <<'P'
import glob, json, collections
res = collections.defaultdict(list)
for f in sorted(glob.glob('gpurun_out/r6w/bench_*.json')):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        res[f.split('/')[-1].split('.')[0]].append((d['ms_per_step'], d['config']['final_loss']))
    except Exception as ex:
        print(f, 'ERR', ex)
for k, v in sorted(res.items()): print(k, v)
P
python tools/step_timeline.py 2>/dev/null | grep -E "image.bwd.end|adam|step.end"
