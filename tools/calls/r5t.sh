cd $GRAFT_REPO_ROOT
O=gpurun_out/r5t; mkdir -p $O
GPTQ_DECODE_C1=1 GPTQ_DECODE_ROWS_WIDE=8 GPTQ_DECODE_ROWS_PAIR=8 timeout 900 python -m pytest tests/test_gpu_batch.py -q -m gpu -x -k "wide_layers or layer_decode" > $O/pytest_env.txt 2>&1; tail -5 $O/pytest_env.txt
timeout 900 python -m pytest tests/test_gpu_batch.py -q -m gpu -x -k "wide_layers" > $O/pytest.txt 2>&1; tail -3 $O/pytest.txt
for V in base new; do
E=""; [ "$V" = "new" ] && E="GPTQ_DECODE_C1=1 GPTQ_DECODE_ROWS_WIDE=8 GPTQ_DECODE_ROWS_PAIR=8"
env $E MS=1,5,8 SHAPES= timeout 300 python tools/bench_layer_decode.py 2>/dev/null | grep -v lm_head | grep "12288\|pair" > $O/layer_$V.txt; cat $O/layer_$V.txt
env $E timeout 600 python - > $O/engine_$V.txt 2>/dev/null <<'PY'
import sys, json
sys.path.insert(0, 'gptq-for-llama_amd')
import torch
from quant.decode import build_random_llama, benchmark_decode_engine
m = build_random_llama('cuda:0')
for B in (1, 5, 8):
    r = benchmark_decode_engine(m, tokens=32, graph=True, batch=B)
    print(json.dumps({'B': B, 'tok_s': r['tokens_per_s'], 'ms_step': 1e3 * (r.get('median_s_per_step') or r.get('median_s_per_token'))}), flush=True)
PY
cat $O/engine_$V.txt
done
