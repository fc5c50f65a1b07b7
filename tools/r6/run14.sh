#!/bin/bash
# nondeterminism hunt: final_loss over repeated runs, one schedule switch at a time
O=gpurun_out/r6o; mkdir -p $O; rm -f $O/*
OLD="TRICOLO_DS_FWD=0 TRICOLO_DS_BWD=0 TRICOLO_PREP_ISSUE=0 TRICOLO_PREP_DGRAD_LATE=0 TRICOLO_IMG_BWD_FIRST=0"
run() { env $1 python bench.py --modes "" --no-cpu-baseline > $O/bench_$2.$3.json 2>> $O/bench.err; }
for rep in 1 2 3 4 5 6 7 8 9 10 11 12; do
  run "$OLD" old $rep
  run "$OLD TRICOLO_DS_FWD=2" fwd2 $rep
  run "$OLD TRICOLO_DS_BWD=2" bwd2 $rep
  run "$OLD TRICOLO_PREP_ISSUE=1" prep1 $rep
  run "$OLD TRICOLO_PREP_DGRAD_LATE=1" late $rep
  run "X=1" new $rep
done
python - <<'P'
import glob, json, collections
res = collections.defaultdict(list)
for f in sorted(glob.glob('gpurun_out/r6o/bench_*.json')):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        res[f.split('/')[-1].split('.')[0]].append((d['ms_per_step'], d['config']['final_loss']))
    except Exception as ex:
        print(f, 'ERR', ex)
for k, v in sorted(res.items()):
    print(k, 'ms', sorted(x[0] for x in v)[len(v)//2], 'losses', collections.Counter(x[1] for x in v))
P
