#!/bin/bash
# greedy generate without a host round trip per token (quant/engine_hook.py _greedy_fast): its test, the model / reference-caller / batch files, tok/s of the bench's generate leg
O=gpurun_out/r7q; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_model.py tests/test_reference_callers.py tests/test_gpu_batch.py -x -q -m gpu -k "generate or engine or hook or callers or process" > $O/pytest.txt 2>&1; tail -4 $O/pytest.txt
python - > $O/generate.txt 2>&1 <<'PY'
import os, sys
sys.path.insert(0, 'gptq-for-llama_amd'); sys.path.insert(0, '.')
import torch
from quant import decode as D, engine_hook as EH
m = D.build_random_llama('cuda:0')
for fast in (True, False, True):
    EH.GREEDY_FAST = fast
    r = D.benchmark_generate(m, prompt_len=16, new_tokens=128)
    print('GREEDY_FAST', fast, {k: r[k] for k in ('tokens_per_s', 's_per_token', 'generated')})
PY
grep GREEDY_FAST $O/generate.txt || tail -5 $O/generate.txt
