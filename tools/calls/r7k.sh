#!/bin/bash
# one loader wave + seven consumers (GPTQ_MMR_NL1 = 2 / 3: chunks in flight 2 / 3-4) against two + six, deeper rings at up to four row tiles; parity first
O=gpurun_out/r7k; mkdir -p $O
for v in 0 2 3; do GPTQ_MMR_NL1=$v timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "loader_consumer_route or stripe_mm_vs_oracle" > $O/pytest_nl1_$v.txt 2>&1; tail -1 $O/pytest_nl1_$v.txt; done
( for v in 0 2 3; do GPTQ_MMR_NL1=$v MS=32,48,64 SHAPES=4096x12288,4096x11008,4096x8192 python tools/bench_mmr.py | sed "s/^/NL1=$v /"; done ) 2>&1 | grep GPTQ_MMR > $O/nl1.txt; cat $O/nl1.txt
