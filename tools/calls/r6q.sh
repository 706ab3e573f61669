#!/bin/bash
# combine + norm launch with its slice loads in flight together: parity (bit-identity test + soak), engine at 9 / 16 rows
O=gpurun_out/r6q; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_batch.py tests/test_gpu_soak.py -x -q -m gpu -k "next_norm or soak_down" > $O/pytest.txt 2>&1; tail -2 $O/pytest.txt
python tools/bench_batches.py 9 16 2>&1 | grep tok_s > $O/engine.txt; GPTQ_NEXT_NORM=0 python tools/bench_batches.py 9 16 2>&1 | grep tok_s >> $O/engine.txt; cat $O/engine.txt
