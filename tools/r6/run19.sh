#!/bin/bash
for m in none mm mem; do BG=$m python tools/r6/text_determinism.py 2>&1 | grep -v amdgpu.ids | tail -2 | cut -c1-600; done
