python -m pytest tests/test_gpu_ops.py -m gpu -x -q -k "bn_fused or stats_acc" 2>&1 | tail -2
for f in 0 1 2; do TRICOLO_BN_FUSED=$f python bench.py --steps 30 --warmup 5 --modes "" --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('BN_FUSED=$f', d['ms_per_step'])"; done
