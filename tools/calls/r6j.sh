#!/bin/bash
# reduce + next norm in the slices' combine launch, TP act-order row shards as fp32 partials: parity, then the engine
O=gpurun_out/r6j; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_batch.py -x -q -m gpu -k "next_norm or engine or generate" > $O/pytest_batch.txt 2>&1; tail -3 $O/pytest_batch.txt
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "row_shards or act_order" > $O/pytest_parity.txt 2>&1; tail -3 $O/pytest_parity.txt
timeout 1500 python -m pytest tests/test_gpu_model.py -x -q -m gpu -k "tensor_parallel" > $O/pytest_tp.txt 2>&1; tail -3 $O/pytest_tp.txt
python tools/bench_batches.py 8 9 16 > $O/engine.txt 2>&1; tail -1 $O/engine.txt | cut -c1-300
GPTQ_NEXT_NORM=0 python tools/bench_batches.py 9 16 > $O/engine_next_norm_0.txt 2>&1; tail -1 $O/engine_next_norm_0.txt | cut -c1-300
