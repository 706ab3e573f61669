cd $GRAFT_REPO_ROOT
O=gpurun_out/r5ac; mkdir -p $O
timeout 100 python - > $O/engine_context.txt 2>/dev/null <<'PY'
import sys, json
sys.path.insert(0, 'gptq-for-llama_amd')
import torch
from quant.decode import build_random_llama, benchmark_decode_engine
m = build_random_llama('cuda:0')
for sp in (0, 500, 1500):
    r = benchmark_decode_engine(m, tokens=32, graph=True, start_pos=sp)
    print(json.dumps({'start_pos': sp, 'tok_s': r['tokens_per_s']}), flush=True)
PY
cat $O/engine_context.txt
