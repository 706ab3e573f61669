#!/bin/bash
# the pair through the set-split loader / consumer kernel as the default: the new route tests + the stripe_mm / fused_mlp files; 113 .. 128 rows with two stripes per workgroup against the spilling three-stripe instance
O=gpurun_out/r7a; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_batch.py -x -q -m gpu -k "stripe_mm or mid_m or layer_decode or wide_layers or stripe_gemm or fused_mlp or pair" > $O/pytest.txt 2>&1; tail -2 $O/pytest.txt
( GPTQ_MMR_PAIR_C8=3 MS=112,128 python tools/bench_pair_mm1.py | sed 's/^/C8=3 /'; MS=16,32,48,64,80,96,112,128 python tools/bench_pair_mm1.py ) 2>&1 | grep "gate/up" > $O/pair_default.txt; cat $O/pair_default.txt
