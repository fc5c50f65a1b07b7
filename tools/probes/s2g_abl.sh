#!/bin/bash
# conv_s2g_kernel timing ablations (probe builds of the library with one part of the chunk loop removed; results are wrong, times are what matters)
cd tricolo_amd/csrc
for abl in BASE NOMMA NOREAD NORING NOSLAB; do
  rm -f conv_s2g.o
  /opt/rocm/bin/hipcc -DS2G_ABL_$abl -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-result -Wno-unused-value -Wno-inline-asm -c conv_s2g.hip -o conv_s2g.o
  /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 *.o -o ../libtricolo_hip.so
  echo "== $abl"; (cd ../..; python tools/conv_layers_bench.py --precision f16 2>&1 | grep -E "l256.c3x3s2|l512.c3x3s2" | cut -c1-75)
done
rm -f conv_s2g.o; make -j8 > /dev/null 2>&1
