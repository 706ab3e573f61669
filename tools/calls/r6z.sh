#!/bin/bash
# the gate | up PAIR at 16 .. 128 rows through the loader / consumer kernel (NS = 2): consumers run both sets (GPTQ_MMR_PAIR_SS=0) or split by set (=1), against the routes before (GPTQ_MMR_PAIR=0)
O=gpurun_out/r6z; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "fused_mlp or pair or stripe_mm" > $O/pytest.txt 2>&1; tail -2 $O/pytest.txt
GPTQ_MMR_PAIR_SS=1 timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "fused_mlp or pair" > $O/pytest_ss1.txt 2>&1; tail -2 $O/pytest_ss1.txt
GPTQ_MMR_PAIR_SS=0 timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "fused_mlp or pair" > $O/pytest_ss0.txt 2>&1; tail -2 $O/pytest_ss0.txt
( GPTQ_MMR_PAIR_SS=0 MS=16,32,48,64,80,96,112,128 python tools/bench_pair_mm1.py | sed 's/^/SS=0 /'; GPTQ_MMR_PAIR_SS=1 MS=16,32,48,64,80,96,112,128 python tools/bench_pair_mm1.py | sed 's/^/SS=1 /'; MS=16,32,48,64,80,96,112,128 python tools/bench_pair_mm1.py ) 2>&1 | grep "gate/up" > $O/pair_ss.txt; cat $O/pair_ss.txt
