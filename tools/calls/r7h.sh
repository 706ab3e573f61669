#!/bin/bash
# K-sliced loader / consumer kernel as the default on long-K one-round shapes: route tests, soak, the stripe_mm files; us per launch on every shape
O=gpurun_out/r7h; mkdir -p $O
timeout 1800 python -m pytest tests/test_gpu_parity.py tests/test_gpu_batch.py tests/test_gpu_soak.py -x -q -m gpu -k "stripe_mm or mid_m or layer_decode or wide_layers or stripe_gemm or fused_mlp or pair or loader_consumer or short_prompt" > $O/pytest.txt 2>&1; tail -2 $O/pytest.txt
MS=16,32,48,64,80,96,112,128 python tools/bench_mmr.py 2>&1 | grep GPTQ_MMR > $O/mmr_default.txt; cat $O/mmr_default.txt
