#!/bin/bash
O=gpurun_out/r6aa; mkdir -p $O; rm -f $O/*
timeout 900 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "overflow_noted or adam" > $O/test_ops.txt 2>&1
tail -30 $O/test_ops.txt | grep -v "^RCCL\|^HIP\|^ROCm" | cut -c1-220
timeout 300 python tools/step_timeline.py 2>/dev/null | grep -E "image.bwd.end|adam|step.end" | sed "s/^/new /"
TRICOLO_GUARD_NOTE=0 timeout 300 python tools/step_timeline.py 2>/dev/null | grep -E "image.bwd.end|adam|step.end" | sed "s/^/old /"
run() { env $1 timeout 600 python bench.py --modes "" --no-cpu-baseline --no-voxel-config5 > $O/bench_$2.$3.json 2>> $O/bench.err; }
for rep in 1 2 3; do run "X=1" new $rep; run "TRICOLO_GUARD_NOTE=0" old $rep; done
python - <<'P'
import glob, json, collections
res = collections.defaultdict(list)
for f in sorted(glob.glob('gpurun_out/r6aa/bench_*.json')):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        res[f.split('/')[-1].split('.')[0]].append((d['ms_per_step'], d['config']['final_loss']))
    except Exception as ex:
        print(f, 'ERR', ex)
for k, v in sorted(res.items()): print(k, v)
P
