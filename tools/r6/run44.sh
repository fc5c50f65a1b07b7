#!/bin/bash
O=gpurun_out/r6ao; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "bn_bwd_pair or bn2d" > $O/test_ops.txt 2>&1
tail -12 $O/test_ops.txt | grep -v "^RCCL\|^HIP\|^ROCm" | cut -c1-220
run() { env $1 timeout 200 python bench.py --modes "" --no-cpu-baseline --no-voxel-config5 > $O/bench_$2.$3.json 2>> $O/bench.err; }
for rep in 1 2 3 4 5 6 7; do run "X=1" pair $rep; run "TRICOLO_BN_PAIR=0" single $rep; done
python - <<'P'
import glob, json, collections
res = collections.defaultdict(list)
for f in sorted(glob.glob('gpurun_out/r6ao/bench_*.json')):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        res[f.split('/')[-1].split('.')[0]].append((d['ms_per_step'], d['config']['final_loss']))
    except Exception as ex:
        print(f, 'ERR', ex)
for k, v in sorted(res.items()): print(k, v)
P
