#!/bin/bash
O=gpurun_out/r6i; mkdir -p $O
run() { env $1 python bench.py --modes "" --no-cpu-baseline > $O/bench_$2.$3.json 2>> $O/bench.err; }
for rep in 1 2 3; do
  run "TRICOLO_DS_ON_TEXT=0" base $rep
  run "TRICOLO_DS_ON_TEXT=1" text $rep
  run "TRICOLO_DS_ON_TEXT=2" vox $rep
  run "TRICOLO_DS_ON_TEXT=1 TRICOLO_TOWER_ORDER=ivt" text_ivt $rep
  run "TRICOLO_DS_ON_TEXT=2 TRICOLO_TOWER_ORDER=ivt" vox_ivt $rep
done
python - <<'P'
import glob, json, collections
res = collections.defaultdict(list)
for f in sorted(glob.glob('gpurun_out/r6i/bench_*.json')):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        res[f.split('/')[-1].split('.')[0]].append(d['ms_per_step'])
    except Exception as ex:
        print(f, 'ERR', ex)
for k, v in sorted(res.items()):
    print(k, v)
P
for e in "TRICOLO_DS_ON_TEXT=1" "TRICOLO_DS_ON_TEXT=2"; do
  echo "== $e"; env $e python tools/step_timeline.py 2>/dev/null | grep -E "step.start|fwd|loss|bwd.start|bwd.end|adam|step.end|gru|layer"
done
