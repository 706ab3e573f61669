#!/bin/bash
O=gpurun_out/r6c; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_batch.py -x -q -m gpu -k "attention or attn or o_proj" > $O/pytest_attn.txt 2>&1
tail -3 $O/pytest_attn.txt
python tools/bench_attn_layout.py 0 127 511 1023 2046 > $O/attn_micro_wgs256.txt 2>&1; cat $O/attn_micro_wgs256.txt | cut -c1-200
GPTQ_ATTN_WGS=128 python tools/bench_attn_layout.py 511 1023 2046 > $O/attn_micro_wgs128.txt 2>&1; cat $O/attn_micro_wgs128.txt | cut -c1-200
( python tools/bench_context.py 1
  GPTQ_ATTN_WGS=128 python tools/bench_context.py 1
  GPTQ_ATTN_TPS_REC=256 python tools/bench_context.py 1
  python tools/bench_context.py 1
  GPTQ_ATTN_WGS=128 python tools/bench_context.py 1
  GPTQ_ATTN_RECORDS=0 python tools/bench_context.py 1 2 4 ) > $O/context.txt 2>&1
grep tok_s $O/context.txt
