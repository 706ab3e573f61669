#!/bin/bash
# sampling generate through a self-feeding sampling graph: its test + the model / batch / caller files; tok/s of the reference script's own call on a 7B-shaped model with FINITE logits (scales x 0.05)
O=gpurun_out/r7v; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_model.py tests/test_reference_callers.py tests/test_gpu_batch.py -x -q -m gpu > $O/pytest.txt 2>&1; tail -3 $O/pytest.txt; grep -n "^E  " $O/pytest.txt | head -8
timeout 600 python - > $O/sample_tok_s.txt 2>&1 <<'PY'
import sys, time
sys.path.insert(0, 'gptq-for-llama_amd'); sys.path.insert(0, '.')
import torch
from quant import decode as D, engine_hook as EH
fill = D.fill_random_quant_
def small(layer, gen):
    fill(layer, gen)
    layer.scales.mul_(0.05)          # the stock random scales overflow fp16 in a 32-layer stack: NaN logits, and torch.multinomial aborts on them (HF's loop too)
D.fill_random_quant_ = small
m = D.build_random_llama('cuda:0')
ids = torch.randint(1, 32000, (1, 16), device='cuda:0')
with torch.no_grad():
    lg = m(ids).logits[0, -1].float()
print('logits finite', bool(torch.isfinite(lg).all()), float(lg.abs().max()), flush=True)
if bool(torch.isfinite(lg).all()):
    def t(n, seed=0):
        torch.manual_seed(seed); torch.cuda.synchronize(); t0 = time.perf_counter()
        with torch.no_grad():
            out = m.generate(ids, do_sample=True, max_new_tokens=n, top_p=0.95, temperature=0.8, eos_token_id=None)
        torch.cuda.synchronize(); return time.perf_counter() - t0, out
    for fast in (True, False, True):
        EH.SAMPLE_FAST = fast
        t(4); t1, _ = t(1); tn, out = t(128)
        print('SAMPLE_FAST', fast, 'generate(do_sample, top_p 0.95, temperature 0.8): tok/s', round(127 / (tn - t1), 1), 'tokens', out[0, 16:24].tolist(), flush=True)
PY
grep "SAMPLE_FAST\|logits finite" $O/sample_tok_s.txt || tail -5 $O/sample_tok_s.txt
