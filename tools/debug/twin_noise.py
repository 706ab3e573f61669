"""tests/test_gpu_model.py::test_tiny_llama_batched_decode_matches_dense_twin at one batch: the drop-in model's logits against the fp16 dense twin (the
test's figure) AND against the twin run in fp32 (the truth both approximate), for A/B of a kernel route (env knobs, e.g. GPTQ_DECODE_MF8=0).
   python tools/debug/twin_noise.py BATCH [SEED ...]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'gptq-for-llama_amd'), os.path.join(ROOT, 'tests')]
import numpy as np, torch
import test_gpu_model as T
from quant import decode as D
batch = int(sys.argv[1])
for seed in [int(a) for a in sys.argv[2:]] or [7]:
    qu = D.build_random_llama(T.DEV, bits=4, groupsize=128, seed=seed, fused=False, **T.TINY)
    ref = T.dense_twin(qu, T.TINY)
    q = D.build_random_llama(T.DEV, bits=4, groupsize=128, seed=seed, fused=True, **T.TINY)
    gen = torch.Generator(device=T.DEV); gen.manual_seed(99 + batch)
    ids = torch.randint(0, T.TINY['vocab_size'], (batch, 8), device=T.DEV, generator=gen)
    a, c = T.run_steps(q, ids, 5), T.run_steps(ref, ids, 5)
    e = T.run_steps(ref.float(), ids, 5)
    scale = np.abs(e).max()
    print({'batch': batch, 'seed': seed, 'knobs': {k: v for k, v in os.environ.items() if k.startswith('GPTQ_')},
           'quant_vs_fp16_twin': round(float(np.abs(a - c).max() / scale), 5), 'quant_vs_fp32_twin': round(float(np.abs(a - e).max() / scale), 5),
           'fp16_twin_vs_fp32_twin': round(float(np.abs(c - e).max() / scale), 5)}, flush=True)
