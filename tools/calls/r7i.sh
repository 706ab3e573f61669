#!/bin/bash
# one row tile (9 .. 16 rows) of down_proj through the K-sliced loader / consumer kernel against the K slices of the wave-shared kernel
O=gpurun_out/r7i; mkdir -p $O
( MS=9,16 SHAPES=11008x4096 python tools/bench_mmr.py | sed 's/^/default /'; for ks in 44 48 24; do GPTQ_MMR_KS=$ks MS=9,16 SHAPES=11008x4096 python tools/bench_mmr.py | sed "s/^/KS=$ks /"; done ) 2>&1 | grep GPTQ_MMR > $O/m16.txt; cat $O/m16.txt
