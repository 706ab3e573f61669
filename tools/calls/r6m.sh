#!/bin/bash
# 5 .. 8 rows of down_proj: the second half of x requested one block early (MF instances of stripe_gemv2p_kernel) against at the phase boundary
O=gpurun_out/r6m; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_model.py -x -q -m gpu -k "row_groups or fused_mlp or long_k or batched_decode_matches" > $O/pytest.txt 2>&1; tail -3 $O/pytest.txt
for e in 1 0 1 0; do GPTQ_X2P_EARLY=$e MS=4,5,8 SHAPES=11008x4096 python tools/bench_layer_decode.py 2>&1 | grep shape | sed "s/^/early=$e /" >> $O/down_2p.txt; done; cat $O/down_2p.txt | cut -c1-200
for e in 1 0; do GPTQ_X2P_EARLY=$e python tools/bench_batches.py 5 8 2>&1 | grep tok_s >> $O/engine.txt; done; cat $O/engine.txt
