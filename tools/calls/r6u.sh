#!/bin/bash
# does the row stride of x (8 KiB: every row of a chunk on the same L2 channel?) bound the 17 .. 128-row kernels -- x with padded rows
O=gpurun_out/r6u; mkdir -p $O
( for pad in 0 128 64 32 16 8; do XPAD=$pad SHAPES2=1 python tools/bench_mmr.py 2>&1 | grep GPTQ_MMR; done ) > $O/xpad.txt; cat $O/xpad.txt
