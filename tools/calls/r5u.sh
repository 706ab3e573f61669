cd $GRAFT_REPO_ROOT
O=gpurun_out/r5u; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_batch.py tests/test_gpu_parity.py -q -m gpu -k "wide_layers or layer_decode or row_groups or stripe_matvec or fused_norm or decode or engine or act_order or matvec" > $O/pytest.txt 2>&1; tail -5 $O/pytest.txt
timeout 600 python - > $O/engine.txt 2>/dev/null <<'PY'
import sys, json
sys.path.insert(0, 'gptq-for-llama_amd')
import torch
from quant.decode import build_random_llama, benchmark_decode_engine
m = build_random_llama('cuda:0')
for B in (1, 2, 4, 5, 8, 16):
    r = benchmark_decode_engine(m, tokens=48, graph=True, batch=B)
    print(json.dumps({'B': B, 'tok_s': r['tokens_per_s'], 'ms_step': 1e3 * (r.get('median_s_per_step') or r.get('median_s_per_token'))}), flush=True)
PY
cat $O/engine.txt
