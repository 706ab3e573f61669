#!/bin/bash
# one-round shapes through the loader / consumer kernel with TWO stripes per workgroup on half the CUs (GPTQ_MMR=2 GPTQ_MMR_C1=2) against one stripe on all (=1) and the default routes
O=gpurun_out/r7m; mkdir -p $O
( MS=32,64,96,128 SHAPES=4096x4096,5120x5120 python tools/bench_mmr.py | sed 's/^/default /'; for c in 1 2; do GPTQ_MMR=2 GPTQ_MMR_KS=0 GPTQ_MMR_C1=$c MS=32,64,96,128 SHAPES=4096x4096,5120x5120 python tools/bench_mmr.py | sed "s/^/C1=$c /"; done ) 2>&1 | grep GPTQ_MMR > $O/c1.txt; cat $O/c1.txt
