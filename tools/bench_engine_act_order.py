#!/usr/bin/env python3
"""DecodeEngine on a LLaMA-7B-shaped 4-bit g128 model WITH act-order g_idx on every linear (BASELINE config 4
flavour; q/k/v and gate/up share their permutations as in a real --act-order checkpoint) vs the trivial-g_idx model.
Protocol of llama.py:385-438 (one token per step, KV cache, sync per step, median)."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'gptq-for-llama_amd')); sys.path.insert(0, ROOT)
import torch
from quant.decode import build_random_llama, benchmark_decode_engine
for act in (False, True):
    model = build_random_llama('cuda:0', act_order=act)
    r = benchmark_decode_engine(model, tokens=48, graph=True)
    print(json.dumps({'act_order': act, 'tokens_per_s': r['tokens_per_s'], 'median_s_per_token': r['median_s_per_token'],
                      'launches_per_token': r['launches_per_token']}), flush=True)
    del model
    torch.cuda.empty_cache()
