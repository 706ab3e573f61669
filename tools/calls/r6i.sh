#!/bin/bash
# MF16 routing (pair plain behind the norm launch): parity of the batch file + soak, layer times, engine at every batch size
O=gpurun_out/r6i; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_batch.py tests/test_gpu_soak.py -x -q -m gpu > $O/pytest_batch.txt 2>&1; tail -3 $O/pytest_batch.txt
MS=8,9,16 python tools/bench_layer_decode.py > $O/layer_decode.txt 2>&1; grep -h shape $O/layer_decode.txt | cut -c1-200
python tools/bench_batches.py 1 4 5 8 9 16 > $O/engine.txt 2>&1; tail -1 $O/engine.txt | cut -c1-300
GPTQ_DECODE_MF16=0 python tools/bench_batches.py 9 16 > $O/engine_mf16_0.txt 2>&1; tail -1 $O/engine_mf16_0.txt | cut -c1-300
