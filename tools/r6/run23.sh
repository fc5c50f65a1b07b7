#!/bin/bash
O=gpurun_out/r6u; mkdir -p $O; rm -f $O/*
run() { env $1 python bench.py $4 --modes "" --no-cpu-baseline > $O/bench_$2.$3.json 2>> $O/bench.err; }
for rep in 1 2 3; do run "X=1" base $rep; run "TRICOLO_CAPTURE_PRIO=1" prio $rep; done
run "X=1" cfg3b32 1 "--config 3 --per-gpu-batch 32"; run "X=1" cfg3b32 2 "--config 3 --per-gpu-batch 32"
python - <<'P'
import glob, json, collections
res = collections.defaultdict(list)
for f in sorted(glob.glob('gpurun_out/r6u/bench_*.json')):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        res[f.split('/')[-1].split('.')[0]].append((d['ms_per_step'], d['config']['final_loss']))
    except Exception as ex:
        print(f, 'ERR', ex)
for k, v in sorted(res.items()): print(k, v)
P
tail -3 $O/bench.err | cut -c1-300
