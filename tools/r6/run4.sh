#!/bin/bash
# round 6, GPU call 4: factorial of the side-branch issue orders (shortcut fwd / bwd, trunk packing), timelines of the best
O=gpurun_out/r6d; mkdir -p $O
run() { env $1 python bench.py --modes "" --no-cpu-baseline > $O/bench_$2.$3.json 2>> $O/bench.err; }
for rep in 1 2; do
 for f in 0 1 2; do for b in 0 1 2; do for p in 0 1 2; do
  run "TRICOLO_DS_FWD=$f TRICOLO_DS_BWD=$b TRICOLO_PREP_ISSUE=$p" f${f}b${b}p${p} $rep
 done; done; done
done
python - <<'P'
import glob, json, collections
res = collections.defaultdict(list)
for f in sorted(glob.glob('gpurun_out/r6d/bench_*.json')):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        res[f.split('/')[-1].split('.')[0]].append(d['ms_per_step'])
    except Exception as ex:
        print(f, 'ERR', ex)
for k, v in sorted(res.items(), key=lambda kv: sum(kv[1]) / len(kv[1])):
    print(k, v)
P
for e in "TRICOLO_DS_FWD=2 TRICOLO_DS_BWD=0 TRICOLO_PREP_ISSUE=0" "TRICOLO_DS_FWD=0 TRICOLO_DS_BWD=0 TRICOLO_PREP_ISSUE=0" "TRICOLO_DS_FWD=2 TRICOLO_DS_BWD=2 TRICOLO_PREP_ISSUE=1"; do
  echo "== $e"; env $e python tools/step_timeline.py 2>/dev/null
done > $O/timelines.txt
cat $O/timelines.txt
