#!/bin/bash
# what the round-6 short-prompt kernels are worth at model level: one forward of [1, T] through the LLaMA-7B-shaped drop-in model, eager wall time and hipGraph GPU time, new routes on / off
O=gpurun_out/r7o; mkdir -p $O
( python tools/bench_short_prompt.py; GPTQ_MMR=0 GPTQ_MMR_PAIR=0 GPTQ_MMR_KS=0 python tools/bench_short_prompt.py ) 2>&1 | grep GPTQ_MMR > $O/short_prompt.txt; cat $O/short_prompt.txt
