#!/bin/bash
O=gpurun_out/r6p; mkdir -p $O; rm -f $O/*
python -m pytest tests -m gpu -x -q -k "halo or bench_plan" > $O/tests.txt 2>&1; grep -E "passed|failed" $O/tests.txt | tail -2
OLD="TRICOLO_DS_FWD=0 TRICOLO_DS_BWD=0 TRICOLO_PREP_ISSUE=0 TRICOLO_PREP_DGRAD_LATE=0 TRICOLO_IMG_BWD_FIRST=0"
run() { env $1 python bench.py --modes "" --no-cpu-baseline > $O/bench_$2.$3.json 2>> $O/bench.err; }
for rep in $(seq 1 16); do
  run "$OLD" old $rep
  run "$OLD TRICOLO_HALO_TOUCH=0 TRICOLO_HALO_XCG=0" oldnotouch $rep
  run "X=1" new $rep
done
python - <<'P'
import glob, json, collections
res = collections.defaultdict(list)
for f in sorted(glob.glob('gpurun_out/r6p/bench_*.json')):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        res[f.split('/')[-1].split('.')[0]].append((d['ms_per_step'], d['config']['final_loss'], d['roofline']['frac']))
    except Exception as ex:
        print(f, 'ERR', ex)
for k, v in sorted(res.items()):
    print(k, 'ms', sorted(x[0] for x in v)[len(v)//2], 'frac', sorted(x[2] for x in v)[len(v)//2], 'losses', collections.Counter(x[1] for x in v))
P
