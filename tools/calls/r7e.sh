#!/bin/bash
# the pair at 113 .. 128 rows of three-stripe shapes as two launches of the four-tile instance (default) against two stripes per workgroup (GPTQ_MMR_PAIR_C8=2); parity of the pair route test
O=gpurun_out/r7e; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "loader_consumer_pair or fused_mlp" > $O/pytest.txt 2>&1; tail -2 $O/pytest.txt
( GPTQ_MMR_PAIR_C8=2 MS=112,120,128 python tools/bench_pair_mm1.py | sed 's/^/C8=2 /'; MS=112,113,120,128 python tools/bench_pair_mm1.py ) 2>&1 | grep "gate/up" > $O/pair_two_launches.txt; cat $O/pair_two_launches.txt
