#!/bin/bash
# (a) stripe_mmr_kernel at five to eight row tiles as (row block, 64-row pass) chunks: parity + us per launch; (b) the decode step as a chain of graphs
O=gpurun_out/r6w; mkdir -p $O
GPTQ_MMR_TMAX=8 timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "stripe_mm or mid_m" > $O/pytest.txt 2>&1; tail -2 $O/pytest.txt
( GPTQ_MMR_TMAX=8 SHAPES2=1 MS=64,80,96,112,128 python tools/bench_mmr.py ) 2>&1 | grep GPTQ_MMR > $O/mmr_tm8.txt; cat $O/mmr_tm8.txt
# (b) ran tools/bench_graph_cuts.py on an experimental DecodeEngine.capture(cuts=...): no gain, not kept -- profiles/r6w_graph_cuts/graph_cuts.txt
