#!/bin/bash
O=gpurun_out/r6ak; mkdir -p $O; rm -f $O/*
run() { env $1 timeout 200 python bench.py --modes "" --no-cpu-baseline --no-voxel-config5 > $O/bench_$2.$3.json 2>> $O/bench.err; }
for rep in 1 2 3; do run "X=1" base $rep; run "TRICOLO_HALO_ROWS=0" halo2d $rep; run "TRICOLO_HALO_PROD=0" noprod $rep; run "TRICOLO_HALO_TM3=0" notm3 $rep; done
python - <<'P'
import glob, json, collections
res = collections.defaultdict(list)
for f in sorted(glob.glob('gpurun_out/r6ak/bench_*.json')):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        res[f.split('/')[-1].split('.')[0]].append((d['ms_per_step'], d['config']['final_loss'], d['roofline']['frac']))
    except Exception as ex:
        print(f, 'ERR', ex)
for k, v in sorted(res.items()): print(k, v)
P
