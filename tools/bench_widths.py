#!/usr/bin/env python3
"""2 / 4 / 8-bit g128 decode matvec on the LLaMA-7B shapes, cold weights (rotation inside one hipGraph): the stripe16 kernel
(default dispatch) next to round 1's rowwave kernel on the checkpoint layout (family='gemv').  M = 1 and M = 4."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'gptq-for-llama_amd')); sys.path.insert(0, ROOT)
import torch
from bench import _time_cold
from quant import quant_linear as QL
dev = 'cuda:0'
gen = torch.Generator(device=dev); gen.manual_seed(0)
GS = 128

def make(bits, K, N):
    G = K // GS
    return (torch.randint(-2**31, 2**31 - 1, (K // 32 * bits, N), dtype=torch.int32, device=dev, generator=gen),
            (torch.rand((G, N), device=dev, generator=gen) * 0.01 + 0.001).half(),
            torch.randint(-2**31, 2**31 - 1, (G, N // 32 * bits), dtype=torch.int32, device=dev, generator=gen),
            (torch.arange(K, device=dev) // GS).to(torch.int32))

for bits in (2, 4, 8):
    for K, N in [(4096, 4096), (4096, 12288), (11008, 4096)]:
        nb = 4 * (K * bits // 32) * N + 4 * (K // GS) * (N * bits // 32) + 2 * (K // GS) * N + 2 * K + 2 * N
        nsets = int(300e6 // nb) + 1
        sets = [make(bits, K, N) for _ in range(nsets)]
        rec = {'bits': bits, 'shape': '%dx%d' % (K, N), 'MB': round(nb / 1e6, 2)}
        for M in (1, 4, 8, 16):
            x = torch.randn((M, K), device=dev, generator=gen).half()
            for fam in (None, 'gemv'):
                us = _time_cold(lambda i: QL.matmul248(x, sets[i][0], sets[i][1], sets[i][2], sets[i][3], bits, 2**bits - 1, family=fam), nsets)
                rec['M%d_%s_us' % (M, 'stripe' if fam is None else 'rowwave')] = round(us, 2)
            if M == 1:
                rec['M1_stripe_GBps'] = round(nb / rec['M1_stripe_us'] / 1e3, 1)
        print(json.dumps(rec), flush=True)
        del sets
        torch.cuda.empty_cache()
