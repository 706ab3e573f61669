#!/bin/bash
# round 6, call b: streaming attention after the fixes -- parity, context sweep, per-kernel times of both modes at two depths
O=gpurun_out/r6b; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_batch.py -x -q -m gpu -k "attention or attn or o_proj" > $O/pytest_attn.txt 2>&1
tail -3 $O/pytest_attn.txt
timeout 900 python -m pytest tests/test_gpu_model.py -x -q -m gpu > $O/pytest_model.txt 2>&1
tail -3 $O/pytest_model.txt
( python tools/bench_context.py 1
  GPTQ_ATTN_RECORDS=0 python tools/bench_context.py 1
  GPTQ_ATTN_NW=8 python tools/bench_context.py 1
  GPTQ_ATTN_TPS_REC=256 python tools/bench_context.py 1
  GPTQ_ATTN_TPS_REC=64 python tools/bench_context.py 1 ) > $O/context.txt 2>&1
grep tok_s $O/context.txt
R=$PWD
cd /tmp && export TMPDIR=/tmp
for mode in 1 0; do for st in 0 1000; do
GPTQ_ATTN_RECORDS=$mode timeout 300 rocprofv3 --kernel-trace --stats -f csv -d $R/$O/prof_${mode}_$st -- python $R/tools/profile_engine.py --start $st > $R/$O/prof_${mode}_$st.txt 2>&1
ST=$(find $R/$O/prof_${mode}_$st -name "*kernel_stats.csv" | head -1); cp "$ST" $R/$O/kernel_stats_records${mode}_start$st.csv; rm -rf $R/$O/prof_${mode}_$st
echo "== records=$mode start=$st"; cut -d, -f1-4 $R/$O/kernel_stats_records${mode}_start$st.csv | head -9
done; done
