"""engine tokens/s at depth (round 6): DecodeEngine(batch = B) with ~ctx tokens of history per row, one engine per batch size.
   python tools/bench_context.py [B ...]      env knobs: GPTQ_ATTN_RECORDS, GPTQ_ATTN_SPLITS, GPTQ_ATTN_TPS_REC, GPTQ_ATTN_TPS, GPTQ_ATTN_NW"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'gptq-for-llama_amd')]
import torch
from quant.decode import build_random_llama, benchmark_decode_engine_context

model = build_random_llama('cuda:0')
knobs = {k: v for k, v in os.environ.items() if k.startswith('GPTQ_ATTN')}
for B in [int(a) for a in sys.argv[1:]] or [1]:
    r = benchmark_decode_engine_context(model, contexts=(16, 128, 256, 512, 768, 1024, 1536, 2047), batch=B)
    print(json.dumps({'B': B, 'knobs': knobs, 'attention': r['attention'], 'tok_s': {k: v['tokens_per_s'] for k, v in r.items() if k.startswith('ctx')}}), flush=True)
