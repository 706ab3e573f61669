"""engine tokens/s per batch size (DecodeEngine(batch = B), hipGraph replay, llama.py:385-438 protocol from an empty cache)
   python tools/bench_batches.py [B ...]"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'gptq-for-llama_amd')]
import torch
from quant.decode import build_random_llama, benchmark_decode_engine
model = build_random_llama('cuda:0')
knobs = {k: v for k, v in os.environ.items() if k.startswith('GPTQ_')}
out = {}
for B in [int(a) for a in sys.argv[1:]] or [1, 2, 4, 5, 8, 16]:
    torch.cuda.empty_cache()
    r = benchmark_decode_engine(model, tokens=32 if B > 1 else 64, graph=True, batch=B)
    out['b%d' % B] = r['tokens_per_s']
print(json.dumps({'knobs': knobs, 'tok_s': out}), flush=True)
