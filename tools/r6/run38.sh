#!/bin/bash
O=gpurun_out/r6ai; mkdir -p $O; rm -f $O/*
timeout 600 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "two_addends or bn2d" > $O/test_ops.txt 2>&1
tail -30 $O/test_ops.txt | grep -v "^RCCL\|^HIP\|^ROCm" | cut -c1-220
run() { env $1 timeout 200 python bench.py --modes "" --no-cpu-baseline --no-voxel-config5 > $O/bench_$2.$3.json 2>> $O/bench.err; }
for rep in 1 2 3; do run "X=1" split1 $rep; run "TRICOLO_DX_SPLIT=0" off $rep; run "TRICOLO_DX_SPLIT=2" split2 $rep; done
python - <<'P'
import glob, json, collections
res = collections.defaultdict(list)
for f in sorted(glob.glob('gpurun_out/r6ai/bench_*.json')):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        res[f.split('/')[-1].split('.')[0]].append((d['ms_per_step'], d['config']['final_loss']))
    except Exception as ex:
        print(f, 'ERR', ex)
for k, v in sorted(res.items()): print(k, v)
P
tail -3 $O/bench.err | cut -c1-300
TRICOLO_FINE_STAMPS=1 timeout 200 python tools/step_timeline.py 2>/dev/null | grep -E "c512.b2|c256.b1.bn2|c256.b0|c128.b2|c64.b1.bn2|step.end" | sed "s/^/split1 /"
