for a in 0 1 4 16 5; do echo "== ABL $a"; TRICOLO_HALO_ABL=$a python tools/conv_layers_bench.py --precision f16 --only resnet 2>&1 | grep -E "c3x3s1 x3" | cut -c1-75; done
