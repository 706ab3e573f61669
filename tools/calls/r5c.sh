# round 5, call 3: B = 16 mismatch hunt, MFMA LM head, decode-kernel row groups on wide shapes (A/B knobs)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5c; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_batch.py -q -x -k "layer_decode or dense_matmat or add_rows or attention" 2>&1 | tail -25 > $O/pytest_ops.txt; tail -8 $O/pytest_ops.txt
timeout 300 python tools/debug_b16.py > $O/debug_b16.txt 2>&1; grep -v Warning $O/debug_b16.txt | tail -14
timeout 600 python -m pytest tests/test_gpu_batch.py tests/test_gpu_model.py -q -k "engine or generate or hook or release or padded" 2>&1 | tail -25 > $O/pytest_engine.txt; tail -10 $O/pytest_engine.txt
MS=1,4,5,8,16 timeout 300 python tools/bench_layer_decode.py > $O/layer_decode_default.txt 2>/dev/null; grep lm_head $O/layer_decode_default.txt
GPTQ_LM_HEAD_MFMA_MIN_ROWS=1 MS=1,2,4 SHAPES=4096x4096 timeout 200 python tools/bench_layer_decode.py 2>/dev/null | grep lm_head > $O/lm_head_mfma_small.txt; cat $O/lm_head_mfma_small.txt
GPTQ_DECODE_ROWS_PAIR=16 GPTQ_DECODE_ROWS_WIDE=16 MS=5,8,16 timeout 300 python tools/bench_layer_decode.py 2>/dev/null | grep -v lm_head > $O/layer_decode_rows16.txt
echo "--- default"; grep -v lm_head $O/layer_decode_default.txt | grep -E '"M": (5|8|16)'
echo "--- decode kernel row groups forced on wide shapes"; cat $O/layer_decode_rows16.txt
