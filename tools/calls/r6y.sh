#!/bin/bash
# stripe_mmr_kernel, five to eight row tiles: both passes per consumer against row halves (GPTQ_MMR_NH); parity of the route test on both
O=gpurun_out/r6y; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_batch.py -x -q -m gpu -k "stripe_mm or mid_m or layer_decode or wide_layers or stripe_gemm" > $O/pytest.txt 2>&1; tail -2 $O/pytest.txt
GPTQ_MMR_NH=2 timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "loader_consumer or stripe_mm_vs_oracle" > $O/pytest_nh2.txt 2>&1; tail -2 $O/pytest_nh2.txt
( MS=80,96,112,128 SHAPES=4096x12288,4096x11008,4096x8192 python tools/bench_mmr.py; GPTQ_MMR_NH=1 MS=80,96,112,128 SHAPES=4096x12288,4096x11008,4096x8192 python tools/bench_mmr.py; GPTQ_MMR_NH=2 MS=80,96,112,128 SHAPES=4096x12288,4096x11008,4096x8192 python tools/bench_mmr.py ) 2>&1 | grep GPTQ_MMR > $O/mmr_nh.txt; cat $O/mmr_nh.txt
