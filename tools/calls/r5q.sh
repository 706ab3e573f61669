# round 5: every row block of a short K requested before the staging phase of a decode batch (GPTQ_DECODE_DU_DEEP) -- A/B
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5q; mkdir -p $O
for D in 0 1; do
GPTQ_DECODE_DU_DEEP=$D MS=2,4,8 timeout 300 python tools/bench_layer_decode.py 2>/dev/null | grep -v lm_head > $O/layer_du$D.txt; cat $O/layer_du$D.txt
GPTQ_DECODE_DU_DEEP=$D timeout 600 python - > $O/engine_du$D.txt 2>/dev/null <<'PY'
import sys, json
sys.path.insert(0, 'gptq-for-llama_amd')
import torch
from quant.decode import build_random_llama, benchmark_decode_engine
m = build_random_llama('cuda:0')
for B in (2, 4, 8):
    r = benchmark_decode_engine(m, tokens=32, graph=True, batch=B)
    print(json.dumps({'B': B, 'tok_s': r['tokens_per_s'], 'ms_step': 1e3 * (r.get('median_s_per_step') or r.get('median_s_per_token'))}), flush=True)
PY
cat $O/engine_du$D.txt
done
