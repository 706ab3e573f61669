#!/bin/bash
# stripe_mmr_kernel at five to eight row tiles (two passes over a 32-KiB-per-64-rows chunk on the same unpacked B fragments): parity, then us per launch against the sliced tile GEMM
O=gpurun_out/r6v; mkdir -p $O
GPTQ_MMR_TMAX=8 timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "stripe_mm or mid_m" > $O/pytest.txt 2>&1; tail -2 $O/pytest.txt
( MS=64,80,96,112,128 python tools/bench_mmr.py; GPTQ_MMR_TMAX=8 MS=64,80,96,112,128 python tools/bench_mmr.py ) 2>&1 | grep GPTQ_MMR > $O/mmr_tm8.txt; cat $O/mmr_tm8.txt
