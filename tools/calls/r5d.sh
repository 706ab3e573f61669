# round 5, call 4: pair kernel with C stripes per workgroup (5..16 rows), who is right at B = 16, engine tok/s per batch
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5d; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_batch.py -q 2>&1 | tail -25 > $O/pytest_batch.txt; tail -6 $O/pytest_batch.txt
timeout 600 python -m pytest tests/test_gpu_parity.py -q -k "stripe_mm or fused_mlp or row_groups or small_batch" 2>&1 | tail -8 > $O/pytest_mm.txt; tail -4 $O/pytest_mm.txt
timeout 300 python tools/debug_b16.py 2>&1 | grep -v Warn | grep "twin\|chain" > $O/debug_b16.txt; cat $O/debug_b16.txt
MS=5,8,16 timeout 300 python - > $O/pair.txt 2>/dev/null <<'PY'
import os, sys, json, subprocess
for env in ({}, {'GPTQ_MM3C_PAIR_PF': '1'}, {'GPTQ_MM3C_PAIR': '0'}):
    e = dict(os.environ); e.update(env)
    out = subprocess.run([sys.executable, 'tools/bench_layer_decode.py'], env=e, capture_output=True, text=True).stdout
    for l in out.splitlines():
        if 'pair' in l: print(json.dumps(env), l)
PY
cat $O/pair.txt
timeout 600 python - > $O/engine_batches.txt 2>/dev/null <<'PY'
import sys, json
sys.path.insert(0, 'gptq-for-llama_amd')
import torch
from quant.decode import build_random_llama, benchmark_decode_engine
m = build_random_llama('cuda:0')
for B in (1, 4, 8, 16):
    r = benchmark_decode_engine(m, tokens=32, graph=True, batch=B)
    print(json.dumps({'B': B, 'tok_s': r['tokens_per_s'], 'ms_step': 1e3 * (r.get('median_s_per_step') or r.get('median_s_per_token'))}), flush=True)
PY
cat $O/engine_batches.txt
