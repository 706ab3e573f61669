#!/bin/bash
# batched greedy generate through the self-feeding greedy graph: the model / batch / reference-caller files in full, tok/s of the bench's batch-4 left-padded generate leg
O=gpurun_out/r7s; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_model.py tests/test_reference_callers.py tests/test_gpu_batch.py -x -q -m gpu > $O/pytest.txt 2>&1; tail -2 $O/pytest.txt
python - > $O/generate_b4.txt 2>&1 <<'PY'
import os, sys
sys.path.insert(0, 'gptq-for-llama_amd'); sys.path.insert(0, '.')
import torch
from quant import decode as D, engine_hook as EH
m = D.build_random_llama('cuda:0')
for fast in (True, False, True):
    EH.GREEDY_FAST = fast
    for b in (4, 16):
        r = D.benchmark_generate(m, prompt_len=16, new_tokens=64, batch=b, left_pad=True)
        print('GREEDY_FAST', fast, 'batch', b, {k: r[k] for k in ('tokens_per_s', 's_per_step', 'generated')})
PY
grep GREEDY_FAST $O/generate_b4.txt || tail -5 $O/generate_b4.txt
