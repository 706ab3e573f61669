#!/bin/bash
# round 6, call a: streaming attention -- probe, parity, context sweep with the knobs
mkdir -p gpurun_out/r6a
tools/probes/bufrange > gpurun_out/r6a/bufrange.txt 2>&1
timeout 1500 python -m pytest tests/test_gpu_batch.py -x -q -m gpu -k "attention or attn or o_proj" > gpurun_out/r6a/pytest_attn.txt 2>&1
tail -5 gpurun_out/r6a/pytest_attn.txt
timeout 900 python -m pytest tests/test_gpu_model.py -x -q -m gpu > gpurun_out/r6a/pytest_model.txt 2>&1
tail -3 gpurun_out/r6a/pytest_model.txt
( python tools/bench_context.py 1
  GPTQ_ATTN_RECORDS=0 python tools/bench_context.py 1
  GPTQ_ATTN_NW=8 python tools/bench_context.py 1
  GPTQ_ATTN_TPS_REC=256 python tools/bench_context.py 1
  GPTQ_ATTN_RECORDS=0 GPTQ_ATTN_TPS=2048 python tools/bench_context.py 1
  python tools/bench_context.py 2 4 16 ) > gpurun_out/r6a/context.txt 2>&1
cat gpurun_out/r6a/bufrange.txt; grep tok_s gpurun_out/r6a/context.txt
