python tools/step_timeline.py --config 2 2>&1 | tail -14
python bench.py --config 2 --steps 20 --warmup 5 --modes "" --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('config 2', d['ms_per_step'], d['value'])"
TRICOLO_OVERLAP=0 python bench.py --config 2 --steps 20 --warmup 5 --modes "" --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('config 2 serial', d['ms_per_step'], d['value'])"
python -m pytest tests/test_gpu_modules.py -m gpu -x -q -k "not heldout" 2>&1 | grep -E "passed|failed|Error" | tail -3
