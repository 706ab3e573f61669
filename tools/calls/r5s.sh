cd $GRAFT_REPO_ROOT
O=gpurun_out/r5s; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_batch.py -q -m gpu -x -k "wide_layers or layer_decode" > $O/pytest.txt 2>&1; tail -15 $O/pytest.txt
for C in 0 -1; do
E=""; [ "$C" = "0" ] && E="GPTQ_DECODE_C=0"
env $E MS=2,4 timeout 300 python tools/bench_layer_decode.py 2>/dev/null | grep -v lm_head > $O/layer_c$C.txt; cat $O/layer_c$C.txt
env $E timeout 600 python - > $O/engine_c$C.txt 2>/dev/null <<'PY'
import sys, json
sys.path.insert(0, 'gptq-for-llama_amd')
import torch
from quant.decode import build_random_llama, benchmark_decode_engine
m = build_random_llama('cuda:0')
for B in (2, 4):
    r = benchmark_decode_engine(m, tokens=32, graph=True, batch=B)
    print(json.dumps({'B': B, 'tok_s': r['tokens_per_s'], 'ms_step': 1e3 * (r.get('median_s_per_step') or r.get('median_s_per_token'))}), flush=True)
PY
cat $O/engine_c$C.txt
done
