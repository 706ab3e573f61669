#!/bin/bash
# stripe_mmr_kernel as the default route for 17 .. 112 rows (.. 128 with two stripes per workgroup) on the wide shapes: parity, then us per launch on every shape
O=gpurun_out/r6x; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_batch.py -x -q -m gpu -k "stripe_mm or mid_m or layer_decode or wide_layers or stripe_gemm" > $O/pytest.txt 2>&1; tail -2 $O/pytest.txt
( MS=16,32,48,64,80,96,112,128 python tools/bench_mmr.py; GPTQ_MMR=0 MS=80,96,112,128 SHAPES=4096x8192,8192x8192 python tools/bench_mmr.py; MS=80,96,112,128 SHAPES=4096x8192,8192x8192 python tools/bench_mmr.py ) 2>&1 | grep GPTQ_MMR > $O/mmr_default.txt; cat $O/mmr_default.txt
