#!/bin/bash
# weights two row blocks ahead in the loader / consumer kernel (PFD = 2) against one (GPTQ_MMR_PFD=1): parity, single sets and the pair
O=gpurun_out/r7c; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_batch.py -x -q -m gpu -k "stripe_mm or mid_m or layer_decode or wide_layers or stripe_gemm or fused_mlp or pair" > $O/pytest.txt 2>&1; tail -2 $O/pytest.txt
( GPTQ_MMR_PFD=1 GPTQ_MMR=2 MS=16,32,48,64,80,96,112,128 python tools/bench_mmr.py | sed 's/^/PFD=1 MMR=2 /'; GPTQ_MMR=2 MS=16,32,48,64,80,96,112,128 python tools/bench_mmr.py | sed 's/^/PFD=2 MMR=2 /'; MS=16,32,48,64,80,96,112,128 python tools/bench_mmr.py ) 2>&1 | grep GPTQ_MMR > $O/mmr_pfd.txt; cat $O/mmr_pfd.txt
( GPTQ_MMR_PFD=1 MS=16,32,64,96,128 python tools/bench_pair_mm1.py | sed 's/^/PFD=1 /'; MS=16,32,48,64,80,96,112,128 python tools/bench_pair_mm1.py ) 2>&1 | grep "gate/up" > $O/pair_pfd.txt; cat $O/pair_pfd.txt
