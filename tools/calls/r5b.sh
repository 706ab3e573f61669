# round 5, call 2: batched decode -- tests, engine tok/s per batch, per-kernel stats of a B = 4 and a B = 16 step
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5b; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_batch.py -q 2>&1 | tail -40 > $O/pytest_batch.txt; tail -15 $O/pytest_batch.txt
timeout 1200 python -m pytest tests -m gpu -q --deselect tests/test_gpu_batch.py 2>&1 | tail -40 > $O/pytest_gpu.txt; tail -8 $O/pytest_gpu.txt
timeout 600 python - > $O/engine_batches.txt 2>$O/engine_batches.err <<'PY'
import sys, os, json
sys.path.insert(0, 'gptq-for-llama_amd')
import torch
from quant.decode import build_random_llama, benchmark_decode_engine, benchmark_generate
m = build_random_llama('cuda:0')
for B in (1, 2, 3, 4, 5, 8, 12, 16):
    r = benchmark_decode_engine(m, tokens=32, graph=True, batch=B)
    print(json.dumps({'B': B, 'tok_s': r['tokens_per_s'], 'ms_step': 1e3 * (r.get('median_s_per_step') or r.get('median_s_per_token')), 'MiB': r['max_memory_MiB']}), flush=True)
for B in (4, 16):
    r = benchmark_decode_engine(m, tokens=32, graph=True, batch=B, fuse_norm=False)
    print(json.dumps({'B': B, 'unfused_norm': True, 'tok_s': r['tokens_per_s']}), flush=True)
print(json.dumps(benchmark_generate(m, batch=4, left_pad=True, new_tokens=64)), flush=True)
print(json.dumps(benchmark_generate(m, batch=1, new_tokens=64)), flush=True)
PY
cat $O/engine_batches.txt; tail -3 $O/engine_batches.err
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for B in 4 16; do
timeout 300 rocprofv3 --kernel-trace --stats -f csv -d $R/$O/prof_b$B -- python $R/tools/profile_engine.py --batch $B > $R/$O/prof_b$B.txt 2>&1
ST=$(find $R/$O/prof_b$B -name "*kernel_stats.csv" | head -1); cp "$ST" $R/$O/decode_engine_b${B}_kernel_stats.csv
head -14 $R/$O/decode_engine_b${B}_kernel_stats.csv | cut -c1-150
rm -rf $R/$O/prof_b$B
done
