#!/bin/bash
# A/B of the culling scans with ten times their slack (make EXTRA_HIPFLAGS="-DPT_BOX_SLACK=1.0e-3f -DPT_SCREEN_MARGIN=0x1.0p-13f" BUILD=build_exp LIBDIR=build_exp/lib build_exp/lib/libptgpu.so first)
run() { python bench.py --workload c5 --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1 c5', round(d['ms_per_step'],2), 'default', round(d['config']['primary_share']['ms_per_step'],2))"; python tools/mesh_scaling.py 2>/dev/null | grep "199712"; }
run base
tools/exp_lib.sh build_exp/lib/libptgpu.so -- bash -c "$(declare -f run); run wide; python tools/needle_probe.py 2 0.05,0.03,0.02,0.015,0.011 24 | grep sin"
