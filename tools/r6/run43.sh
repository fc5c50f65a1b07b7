#!/bin/bash
O=gpurun_out/r6an; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 500 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/trace -- python3 $GRAFT_REPO_ROOT/bench.py --modes "" --no-cpu-baseline --no-voxel-config5 --steps 10 --warmup 3 --windows 1 > $GRAFT_REPO_ROOT/$O/trace_bench.json 2> $GRAFT_REPO_ROOT/$O/trace_bench.err
cd $GRAFT_REPO_ROOT
python tools/trace_step.py $O/trace --shortest > $O/trace_step.txt 2>&1
find $O/trace -name "*.csv" -size +1M -delete
wc -l $O/trace_step.txt
run() { env $1 timeout 200 python bench.py --modes "" --no-cpu-baseline --no-voxel-config5 > $O/bench_$2.$3.json 2>> $O/bench.err; }
for rep in 1 2 3; do run "X=1" pair $rep; run "TRICOLO_BN_PAIR=0" single $rep; done
python - <<'P'
import glob, json, collections
res = collections.defaultdict(list)
for f in sorted(glob.glob('gpurun_out/r6an/bench_*.json')):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        res[f.split('/')[-1].split('.')[0]].append((d['ms_per_step'], d['config']['final_loss']))
    except Exception as ex:
        print(f, 'ERR', ex)
for k, v in sorted(res.items()): print(k, v)
P
