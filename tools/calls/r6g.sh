#!/bin/bash
# soak: reduced launch count first to see how long one instance takes, then the full count
O=gpurun_out/r6g; mkdir -p $O
( time GPTQ_SOAK_LAUNCHES=2000 timeout 900 python -m pytest tests/test_gpu_soak.py -x -q -m gpu ) > $O/soak_2000.txt 2>&1; tail -8 $O/soak_2000.txt
( time timeout 2400 python -m pytest tests/test_gpu_soak.py -x -q -m gpu --durations=10 ) > $O/soak_full.txt 2>&1; tail -20 $O/soak_full.txt
