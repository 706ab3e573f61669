#!/bin/bash
# loader / consumer small-batch kernel (stripe_mmr_kernel): parity, then us per launch against the default routes
O=gpurun_out/r6s; mkdir -p $O
GPTQ_MMR=1 timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "stripe_mm_vs_oracle" > $O/pytest.txt 2>&1; tail -1 $O/pytest.txt
( for cfg in "3 2" "4 1" "2 2" "3 1"; do set -- $cfg; GPTQ_MMR=1 GPTQ_MMR_NL=$1 GPTQ_MMR_INF=$2 MS=32,64 SHAPES2=1 python tools/bench_mmr.py 2>&1 | grep GPTQ_MMR | sed "s/^/NL=$1 INF=$2 /"; done ) > $O/mmr_sweep2.txt; cut -c1-200 $O/mmr_sweep2.txt
( python tools/bench_mmr.py; GPTQ_MMR=1 python tools/bench_mmr.py ) 2>&1 | grep GPTQ_MMR > $O/mmr2.txt; cat $O/mmr2.txt
