#!/bin/bash
# MF16: 9 .. 16 rows in the decode launch -- parity, then us per launch with GPTQ_DECODE_MF16 = 0 / 1 / 2, then the engine at B = 9 / 16
O=gpurun_out/r6h; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_batch.py tests/test_gpu_soak.py -x -q -m gpu > $O/pytest_batch.txt 2>&1; tail -5 $O/pytest_batch.txt
for m in 0 1 2; do GPTQ_DECODE_MF16=$m MS=8,9,12,16 python tools/bench_layer_decode.py > $O/layer_decode_mf16_$m.txt 2>&1; done
grep -h shape $O/layer_decode_mf16_*.txt | cut -c1-200
for m in 0 1 2; do GPTQ_DECODE_MF16=$m python tools/bench_batches.py 9 16 > $O/engine_mf16_$m.txt 2>&1; tail -3 $O/engine_mf16_$m.txt | cut -c1-300; done
