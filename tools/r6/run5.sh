#!/bin/bash
# round 6, GPU call 5: side towers issued from inside the image tower's forward (TRICOLO_SIDE_AT) x tower order
O=gpurun_out/r6e; mkdir -p $O
run() { env $1 python bench.py --modes "" --no-cpu-baseline > $O/bench_$2.$3.json 2>> $O/bench.err; }
for rep in 1 2; do
 for at in none stem pool l1 l2 l3; do for ord in itv ivt; do
  a=$at; [ $at = none ] && a=""
  run "TRICOLO_SIDE_AT=$a TRICOLO_TOWER_ORDER=$ord" ${at}_$ord $rep
 done; done
done
python - <<'P'
import glob, json, collections
res = collections.defaultdict(list)
for f in sorted(glob.glob('gpurun_out/r6e/bench_*.json')):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        res[f.split('/')[-1].split('.')[0]].append(d['ms_per_step'])
    except Exception as ex:
        print(f, 'ERR', ex)
for k, v in sorted(res.items(), key=lambda kv: sum(kv[1]) / len(kv[1])):
    print(k, v)
P
tail -3 $O/bench.err
for e in "TRICOLO_SIDE_AT=" "TRICOLO_SIDE_AT=stem" "TRICOLO_SIDE_AT=l1" "TRICOLO_SIDE_AT=l2"; do
  echo "== $e"; env $e python tools/step_timeline.py 2>/dev/null | grep -E "step.start|fwd.end|loss|bwd.start|bwd.end|adam|step.end|gru|layer"
done > $O/timelines.txt
cat $O/timelines.txt
