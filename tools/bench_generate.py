#!/usr/bin/env python3
"""End-to-end greedy generation on the LLaMA-7B-shaped random model: 128-token prompt + 128 new tokens through
`engine_generate` (HF prefill with the drop-in modules, then one hipGraph replay per token) vs `model.generate`
(HF eager decode with the same modules; what llama_inference.py:109-115 does)."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'gptq-for-llama_amd')); sys.path.insert(0, ROOT)
import torch
from quant.decode import build_random_llama, DecodeEngine, engine_generate
dev = 'cuda:0'
model = build_random_llama(dev)
gen = torch.Generator(device=dev); gen.manual_seed(0)
prompt = torch.randint(0, 32000, (1, 128), device=dev, generator=gen)
NEW = 128
eng = DecodeEngine(model, t_max=2048).capture()
for _ in range(2):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    seq = engine_generate(model, prompt, NEW, engine=eng)
    torch.cuda.synchronize(); t_eng = time.perf_counter() - t0
with torch.no_grad():
    for _ in range(2):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        out = model.generate(prompt, do_sample=False, max_new_tokens=NEW, min_new_tokens=NEW, pad_token_id=0)
        torch.cuda.synchronize(); t_hf = time.perf_counter() - t0
print(json.dumps({'prompt': 128, 'new_tokens': NEW, 'engine_generate_s': round(t_eng, 4), 'engine_new_tokens_per_s': round(NEW / t_eng, 1),
                  'hf_generate_s': round(t_hf, 4), 'hf_new_tokens_per_s': round(NEW / t_hf, 1),
                  'same_first_tokens': int((seq[0, 128:136] == out[0, 128:136]).sum())}))
