#!/bin/bash
O=gpurun_out/r6m; mkdir -p $O; rm -f $O/*
python -m pytest tests -m gpu -x -q -k "halo or bench_plan or conv_16bit" > $O/tests.txt 2>&1; grep -E "passed|failed" $O/tests.txt | tail -2
python tools/conv_layers_bench.py --precision f16 --only resnet --layers "c3x3s1" --no-wgrad 2>&1 | grep -v amdgpu.ids
run() { env $1 python bench.py --modes "" --no-cpu-baseline > $O/bench_$2.$3.json 2>> $O/bench.err; }
for rep in 1 2 3; do run "X=1" base $rep; done
python - <<'P'
import glob, json, collections
for f in sorted(glob.glob('gpurun_out/r6m/bench_*.json')):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        r = d['roofline']
        print(d['ms_per_step'], r['kernel'][:24], r['frac'], r['avg_launch_ms'], r['families']['conv_halo_rows_kernel'], r['families']['conv_halo2d_kernel'])
    except Exception as ex:
        print(f, 'ERR', ex)
P
tail -3 $O/bench.err
