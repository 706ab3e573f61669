# round 5, call 6: released-layer prefill from the image, long-context attention splits, fallback scratch
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5f; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_batch.py -q -k "released or long_context or attention" 2>&1 | tail -15 > $O/pytest.txt; tail -6 $O/pytest.txt
timeout 600 python -m pytest tests/test_gpu_model.py -q -k "engine or release or tensor_parallel_engine_shards" 2>&1 | tail -10 > $O/pytest_model.txt; tail -4 $O/pytest_model.txt
timeout 300 python tools/bench_released_prefill.py > $O/released_prefill.txt 2>/dev/null; cat $O/released_prefill.txt
for LS in 1 0; do
GPTQ_ATTN_LONG_SPLITS=$LS timeout 400 python - 2>/dev/null <<'PY' | tee -a $O/engine_context.txt
import sys, os, json
sys.path.insert(0, 'gptq-for-llama_amd')
import torch
from quant.decode import build_random_llama, benchmark_decode_engine
m = build_random_llama('cuda:0')
for sp in (0, 500, 1000, 1100, 1500, 1900):
    r = benchmark_decode_engine(m, tokens=32, graph=True, start_pos=sp)
    print(json.dumps({'long_splits': os.environ['GPTQ_ATTN_LONG_SPLITS'], 'start_pos': sp, 'tok_s': r['tokens_per_s']}), flush=True)
PY
done
