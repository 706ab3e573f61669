# round 5, call 5: two-phase row groups on long K, tensor-parallel robustness (shard at load, 8 processes, forced eager), engine tok/s
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5e; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_parity.py -q -x -k "long_k or row_groups or small_batch_rows" 2>&1 | tail -12 > $O/pytest_2p.txt; tail -5 $O/pytest_2p.txt
timeout 900 python -m pytest tests/test_gpu_model.py -q -k "tensor_parallel or bench" 2>&1 | tail -40 > $O/pytest_tp.txt; tail -25 $O/pytest_tp.txt
timeout 600 python -m pytest tests/test_gpu_parity.py -q -k "p2p_allreduce or row_sharded" 2>&1 | tail -15 > $O/pytest_p2p.txt; tail -6 $O/pytest_p2p.txt
MS=4,5,8 SHAPES=11008x4096 timeout 200 python tools/bench_layer_decode.py 2>/dev/null | grep -v lm_head > $O/down_2p.txt; cat $O/down_2p.txt
timeout 600 python - > $O/engine_batches.txt 2>/dev/null <<'PY'
import sys, json
sys.path.insert(0, 'gptq-for-llama_amd')
import torch
from quant.decode import build_random_llama, benchmark_decode_engine
m = build_random_llama('cuda:0')
for B in (5, 8, 12, 16):
    r = benchmark_decode_engine(m, tokens=32, graph=True, batch=B)
    print(json.dumps({'B': B, 'tok_s': r['tokens_per_s'], 'ms_step': 1e3 * (r.get('median_s_per_step') or r.get('median_s_per_token'))}), flush=True)
PY
cat $O/engine_batches.txt
