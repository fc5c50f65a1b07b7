python -m pytest tests/test_gpu_ops.py -m gpu -x -q 2>&1 | tail -4
python -m pytest tests/test_gpu_modules.py -m gpu -x -q -k "not heldout" 2>&1 | grep -E "passed|failed|Error" | tail -4
for c in 4 2 5; do for f in 1 0; do TRICOLO_VOXEL_COMPACT=$f python bench.py --config $c --steps 20 --warmup 5 --modes "" --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('config $c COMPACT=$f', d['ms_per_step'])"; done; done
