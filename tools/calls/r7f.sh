#!/bin/bash
# bounded waits in the loader / consumer kernel: the route tests, the new soak of its instances, times unchanged?
O=gpurun_out/r7f; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_soak.py -x -q -m gpu -k "loader_consumer or short_prompt_tiles or fused_mlp or stripe_mm_vs_oracle" > $O/pytest.txt 2>&1; tail -2 $O/pytest.txt
( MS=32,64,96,128 SHAPES2=1 python tools/bench_mmr.py ) 2>&1 | grep GPTQ_MMR > $O/mmr.txt; cat $O/mmr.txt
MS=32,64,96,128 python tools/bench_pair_mm1.py 2>&1 | grep "gate/up" > $O/pair.txt; cat $O/pair.txt
