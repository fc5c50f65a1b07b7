#!/bin/bash
O=gpurun_out/r6t; mkdir -p $O; rm -f $O/*
python -m pytest tests -m gpu -x -q > $O/tests.txt 2>&1; grep -E "passed|failed|error" $O/tests.txt | tail -3
for rep in 1 2 3; do python bench.py --modes "" --no-cpu-baseline > $O/bench.$rep.json 2>> $O/bench.err; done
python - <<'P'
import glob, json
for f in sorted(glob.glob('gpurun_out/r6t/bench.*.json')):
    d = json.loads(open(f).read().strip().splitlines()[-1]); r = d['roofline']
    print(d['ms_per_step'], d['config']['final_loss'], r['kernel'][:24], r['frac'], r['traffic'])
P
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
