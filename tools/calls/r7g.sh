#!/bin/bash
# K slices of the loader / consumer kernel on one-round shapes (GPTQ_MMR_KS = C*10 + S): parity on 11008 x 4096 / 4096^2, us per launch against the default routes
O=gpurun_out/r7g; mkdir -p $O
for ks in 44 22; do GPTQ_MMR_KS=$ks timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "stripe_mm_vs_oracle or stripe_mm_strided or mid_m" > $O/pytest_ks$ks.txt 2>&1; tail -1 $O/pytest_ks$ks.txt; done
( MS=32,48,64,80,96,128 SHAPES=11008x4096,4096x4096 python tools/bench_mmr.py | sed 's/^/KS=0  /'
  for ks in 44 42 24 22; do GPTQ_MMR_KS=$ks MS=32,48,64,80,96,128 SHAPES=11008x4096,4096x4096 python tools/bench_mmr.py | sed "s/^/KS=$ks /"; done ) 2>&1 | grep GPTQ_MMR > $O/mmr_ks.txt; cat $O/mmr_ks.txt
