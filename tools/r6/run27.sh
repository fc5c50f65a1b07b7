#!/bin/bash
O=gpurun_out/r6y; mkdir -p $O; rm -f $O/*
timeout 900 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "overflow_noted or adam" > $O/test_ops.txt 2>&1
tail -30 $O/test_ops.txt | grep -v "^RCCL\|^HIP\|^ROCm" | cut -c1-220
python tools/step_timeline.py > $O/timeline_new.txt 2>/dev/null
TRICOLO_GUARD_NOTE=0 python tools/step_timeline.py > $O/timeline_old.txt 2>/dev/null
paste $O/timeline_new.txt $O/timeline_old.txt | cut -c1-200 | tail -6
run() { env $1 python bench.py --modes "" --no-cpu-baseline --no-voxel-config5 > $O/bench_$2.$3.json 2>> $O/bench.err; }
for rep in 1 2 3; do run "X=1" new $rep; run "TRICOLO_GUARD_NOTE=0" old $rep; done
python - <<'P'
import glob, json, collections
res = collections.defaultdict(list)
for f in sorted(glob.glob('gpurun_out/r6y/bench_*.json')):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        res[f.split('/')[-1].split('.')[0]].append((d['ms_per_step'], d['config']['final_loss']))
    except Exception as ex:
        print(f, 'ERR', ex)
for k, v in sorted(res.items()): print(k, v)
P
cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats -d /root/repo/gpurun_out/r6y/prof -o new -- python3 /root/repo/bench.py --modes "" --no-cpu-baseline --no-voxel-config5 --steps 20 > /dev/null 2>&1
cd /root/repo; f=$(ls gpurun_out/r6y/prof/*/*kernel_stats.csv 2>/dev/null | head -1); [ -z "$f" ] && f=$(find gpurun_out/r6y/prof -name "*kernel_stats.csv" | head -1); grep -i "reduce_grouped\|adam" $f | cut -c1-200
