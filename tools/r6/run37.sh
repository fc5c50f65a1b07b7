#!/bin/bash
timeout 120 python tools/r6/dbg_pair.py 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|amdgpu.ids" | tail -12
