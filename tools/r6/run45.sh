#!/bin/bash
O=gpurun_out/r6ap; mkdir -p $O
run() { env $1 timeout 200 python bench.py --modes "" --no-cpu-baseline --no-voxel-config5 > $O/bench_$2.$3.json 2>> $O/bench.err; }
for rep in 1 2 3 4 5; do run "X=1" base $rep; run "TRICOLO_PREP_DGRAD_POS=1" pos $rep; done
python - <<'P'
import glob, json, collections
res = collections.defaultdict(list)
for f in sorted(glob.glob('gpurun_out/r6ap/bench_*.json')):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        res[f.split('/')[-1].split('.')[0]].append((d['ms_per_step'], d['config']['final_loss']))
    except Exception as ex:
        print(f, 'ERR', ex)
for k, v in sorted(res.items()): print(k, v)
P
tail -2 $O/bench.err | cut -c1-200
TRICOLO_PREP_DGRAD_POS=1 timeout 200 python tools/step_timeline.py 2>/dev/null | grep -E "fwd|loss|bwd.start|heads|step.end" | sed "s/^/pos /"
