python -m pytest tests/test_gpu_ops.py -q -x -k "conv" 2>&1 | tail -5
for r in 0 1; do echo "== ROWS $r"; TRICOLO_HALO_ROWS=$r python tools/conv_layers_bench.py --precision f16 --only resnet 2>&1 | grep -E "c3x3s1 x3" | cut -c1-100; done
for a in 1 4 5; do echo "== ROWS 1 ABL $a"; TRICOLO_HALO_ABL=$a python tools/conv_layers_bench.py --precision f16 --only resnet 2>&1 | grep -E "c3x3s1 x3" | cut -c1-100; done
python bench.py --modes "" --no-cpu-baseline 2>&1 | grep -o "\"ms_per_step\": [0-9.]*" | head -1
