#!/usr/bin/env python3
"""tools/profile_eager_host.py -- where does the HOST time of the module chain go?  cProfile over batch-1 decode steps of a LLaMA-7B-shaped random
model with the decode engine switched off (the path batch > 1 and padded prompts take): top functions by cumulative and by own time."""
import cProfile, io, os, pstats, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'gptq-for-llama_amd')); sys.path.insert(0, ROOT)
import torch
from quant import decode as D
dev = 'cuda:0'
layers = int(os.environ.get('LAYERS', '8'))
model = D.build_random_llama(dev, num_hidden_layers=layers)
model._gptq_engine_disabled = True
from transformers.cache_utils import DynamicCache
B = int(os.environ.get('BATCH', '1'))
ids = torch.randint(0, 32000, (B, 8), device=dev)
cache = DynamicCache(config=model.config)
with torch.no_grad():
    out = model(ids, past_key_values=cache, use_cache=True)
    tok = out.logits[:, -1:].argmax(-1)
    for _ in range(5):
        out = model(tok, past_key_values=cache, use_cache=True)
    torch.cuda.synchronize()
    import time
    t0 = time.perf_counter()
    for _ in range(20):
        out = model(tok, past_key_values=cache, use_cache=True)
    torch.cuda.synchronize()
    print('%d layers, batch %d: %.1f us per layer and step (wall)' % (layers, B, (time.perf_counter() - t0) / 20 / layers * 1e6))
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(20):
        out = model(tok, past_key_values=cache, use_cache=True)
    torch.cuda.synchronize()
    pr.disable()
for key in ('cumulative', 'tottime'):
    s = io.StringIO()
    pstats.Stats(pr, stream=s).sort_stats(key).print_stats(28)
    txt = s.getvalue()
    print('\n'.join(l[-150:] if len(l) > 150 else l for l in txt.splitlines()[4:44]))
