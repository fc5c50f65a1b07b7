#!/bin/bash
O=gpurun_out/r6am; mkdir -p $O; rm -f $O/*
timeout 600 python -m pytest tests -x -q -m gpu -k "gru or bigru or text" > $O/tests.txt 2>&1; tail -3 $O/tests.txt
run() { env $1 timeout 200 python bench.py --modes "" --no-cpu-baseline --no-voxel-config5 > $O/bench_$2.$3.json 2>> $O/bench.err; }
for rep in 1 2 3 4; do run "X=1" new $rep; done
python - <<'P'
import glob, json, collections
res = collections.defaultdict(list)
for f in sorted(glob.glob('gpurun_out/r6am/bench_*.json')):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        res[f.split('/')[-1].split('.')[0]].append((d['ms_per_step'], d['config']['final_loss']))
    except Exception as ex:
        print(f, 'ERR', ex)
for k, v in sorted(res.items()): print(k, v)
P
tail -2 $O/bench.err | cut -c1-200
TRICOLO_FINE_STAMPS=1 timeout 200 python tools/step_timeline.py 2>/dev/null | grep -E "text\.|image.fwd.end|loss.fwd|step.end"
timeout 200 python tools/step_timeline.py 2>/dev/null | grep -E "fwd.end|loss.fwd|step.end"
timeout 200 python tools/kernel_times.py 2>/dev/null | grep -i "gru"
timeout 200 python bench.py --config 2 --modes "" --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('config2', d['ms_per_step'], d['value'])"
