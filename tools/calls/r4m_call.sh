#!/bin/bash
# round 4, last kernel change: 3-bit act-order on the group-sorted image + the magic-offset table of the 2-/3-bit kernels replaced by selects
mkdir -p gpurun_out/r4m
for s in 1 0; do echo "BITS=3 SORT=$s"; BITS=3 SORT=$s python tools/act_order_probe.py 2>&1 | grep " us"; done > gpurun_out/r4m/act_order_probe_w3.txt
python -m pytest tests -q -m gpu -x -k "config4" 2>&1 | tail -2 > gpurun_out/r4m/config4_tests.txt
python - > gpurun_out/r4m/config4.json 2>gpurun_out/r4m/config4.err <<'PY'
import json, sys, torch
sys.argv = ['bench.py']
import bench
print(json.dumps(bench.config4_leg(torch.device('cuda:0')), indent=1))
PY
tail -3 gpurun_out/r4m/config4_tests.txt; cat gpurun_out/r4m/act_order_probe_w3.txt; grep -c us_per_launch gpurun_out/r4m/config4.json
