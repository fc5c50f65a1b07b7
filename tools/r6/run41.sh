#!/bin/bash
O=gpurun_out/r6al; mkdir -p $O; rm -f $O/*
run() { env $1 timeout 200 python bench.py --modes "" --no-cpu-baseline --no-voxel-config5 > $O/bench_$2.$3.json 2>> $O/bench.err; }
for rep in 1 2 3; do run "X=1" base $rep; run "TRICOLO_MERGE_STREAMS=prep=ds" prepds $rep; run "TRICOLO_MERGE_STREAMS=prep=side" prepside $rep; run "TRICOLO_MERGE_STREAMS=prep=ds,ds=side" all $rep; done
python - <<'P'
import glob, json, collections
res = collections.defaultdict(list)
for f in sorted(glob.glob('gpurun_out/r6al/bench_*.json')):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        res[f.split('/')[-1].split('.')[0]].append((d['ms_per_step'], d['config']['final_loss']))
    except Exception as ex:
        print(f, 'ERR', ex)
for k, v in sorted(res.items()): print(k, v)
P
tail -2 $O/bench.err | cut -c1-200
for v in "X=1" "TRICOLO_MERGE_STREAMS=prep=ds" "TRICOLO_MERGE_STREAMS=prep=side"; do
env $v timeout 200 python tools/step_timeline.py 2>/dev/null | grep -E "fwd.end|loss.fwd|step.end" | sed "s/^/$v /"
done
