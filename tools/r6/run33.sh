#!/bin/bash
O=gpurun_out/r6ae; mkdir -p $O; rm -f $O/*
timeout 600 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "head_chain or linear_small or l2norm or avgpool" > $O/test_ops.txt 2>&1
tail -40 $O/test_ops.txt | grep -v "^RCCL\|^HIP\|^ROCm" | cut -c1-220
run() { env $1 timeout 600 python bench.py --modes "" --no-cpu-baseline --no-voxel-config5 > $O/bench_$2.$3.json 2>> $O/bench.err; }
for rep in 1 2 3; do run "X=1" chain $rep; run "TRICOLO_HEAD_CHAIN=0" single $rep; done
python - <<'P'
import glob, json, collections
res = collections.defaultdict(list)
for f in sorted(glob.glob('gpurun_out/r6ae/bench_*.json')):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        res[f.split('/')[-1].split('.')[0]].append((d['ms_per_step'], d['config']['final_loss']))
    except Exception as ex:
        print(f, 'ERR', ex)
for k, v in sorted(res.items()): print(k, v)
P
tail -5 $O/bench.err | cut -c1-300
timeout 300 python tools/step_timeline.py 2>/dev/null | grep -E "image.fwd.layer4|image.fwd.end|loss.fwd|image.bwd.start|heads|step.end" | sed "s/^/chain /"
TRICOLO_HEAD_CHAIN=0 timeout 300 python tools/step_timeline.py 2>/dev/null | grep -E "image.fwd.layer4|image.fwd.end|loss.fwd|image.bwd.start|heads|step.end" | sed "s/^/single /"
