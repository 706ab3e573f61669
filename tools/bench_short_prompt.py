#!/usr/bin/env python3
"""A short prompt through the drop-in model (eager module chain, one forward of [1, T] with a fresh cache): wall time per forward and the GPU time
of the same forward replayed from a hipGraph (the kernels alone) -- run once per GPTQ_MMR* setting.   TS=32,64,128 python tools/bench_short_prompt.py"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'gptq-for-llama_amd')); sys.path.insert(0, ROOT)
import numpy as np
import torch
from transformers.cache_utils import DynamicCache
from quant import decode as D
dev = 'cuda:0'
model = D.build_random_llama(dev)
out = {'GPTQ_MMR': os.environ.get('GPTQ_MMR', ''), 'GPTQ_MMR_PAIR': os.environ.get('GPTQ_MMR_PAIR', ''), 'GPTQ_MMR_KS': os.environ.get('GPTQ_MMR_KS', '')}
for T in [int(v) for v in os.environ.get('TS', '32,64,128').split(',')]:
    ids = torch.randint(0, model.config.vocab_size, (1, T), device=dev)
    def fwd():
        with torch.no_grad():
            return model(ids, past_key_values=DynamicCache(config=model.config), use_cache=True).logits
    for _ in range(3): fwd()
    torch.cuda.synchronize()
    ts = []
    for _ in range(10):
        torch.cuda.synchronize(); t0 = time.perf_counter(); fwd(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    row = {'eager_ms': round(float(np.median(ts)) * 1e3, 3)}
    try:
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            fwd()
        g.replay(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): g.replay()
        e1.record(); torch.cuda.synchronize()
        row['graph_ms'] = round(e0.elapsed_time(e1) / 10, 3)
    except Exception as ex:
        row['graph_ms'] = 'capture failed: %s' % type(ex).__name__
    out['T%d' % T] = row
print(json.dumps(out), flush=True)
