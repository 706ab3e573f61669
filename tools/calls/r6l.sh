#!/bin/bash
# the batch-8 twin test sits at 1.37e-2 against a 1.2e-2 bar since 5..8 rows go through the 16x16x16 inner product: noise or error?
O=gpurun_out/r6l; mkdir -p $O
( for b in 5 8; do python tools/debug/twin_noise.py $b 7 8 9; GPTQ_DECODE_MF8=0 python tools/debug/twin_noise.py $b 7 8 9; done ) 2>&1 | grep batch > $O/twin_noise.txt; cat $O/twin_noise.txt
timeout 2400 python -m pytest tests -q -m gpu --deselect "tests/test_gpu_model.py::test_tiny_llama_batched_decode_matches_dense_twin" > $O/pytest_gpu.txt 2>&1; tail -5 $O/pytest_gpu.txt
