cd $GRAFT_REPO_ROOT
O=gpurun_out/r5r; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_batch.py -q -m gpu -k "norm or engine" > $O/pytest.txt 2>&1; tail -4 $O/pytest.txt
timeout 600 python - > $O/engine.txt 2>/dev/null <<'PY'
import sys, json
sys.path.insert(0, 'gptq-for-llama_amd')
import torch
from quant.decode import build_random_llama, benchmark_decode_engine
m = build_random_llama('cuda:0')
for B in (1, 4, 16):
    r = benchmark_decode_engine(m, tokens=32, graph=True, batch=B)
    print(json.dumps({'B': B, 'tok_s': r['tokens_per_s'], 'ms_step': 1e3 * (r.get('median_s_per_step') or r.get('median_s_per_token'))}), flush=True)
PY
cat $O/engine.txt
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -f csv -d $GRAFT_REPO_ROOT/$O/prof16 -- python $GRAFT_REPO_ROOT/tools/profile_engine.py --batch 16 > $GRAFT_REPO_ROOT/$O/prof16.txt 2>&1
cd $GRAFT_REPO_ROOT; ST=$(find $O/prof16 -name "*kernel_stats.csv" | head -1); cp "$ST" $O/decode_engine_b16_kernel_stats.csv; rm -rf $O/prof16; head -12 $O/decode_engine_b16_kernel_stats.csv | cut -c1-160
