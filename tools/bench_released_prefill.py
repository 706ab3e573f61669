#!/usr/bin/env python3
"""memory mode at prompt sizes: gptq_layer_forward(M) on a layer with both copies against the same layer released (image only: W^T straight
from the image, stripe_dequant_t_kernel).  us per call, events over REPS calls; MS=2600,4096,16384."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'gptq-for-llama_amd')); sys.path.insert(0, ROOT)
import torch
from quant.layer import PreparedLayer
dev = torch.device('cuda:0'); gen = torch.Generator(device=dev); gen.manual_seed(0)
MS = [int(v) for v in os.environ.get('MS', '2600,4096,16384').split(',')]
REPS = int(os.environ.get('REPS', '10'))


def rand_set(K, N, G):
    return (torch.randint(-2**31, 2**31 - 1, (K // 8, N), dtype=torch.int32, device=dev, generator=gen), (torch.rand((G, N), device=dev, generator=gen) * 0.01 + 0.001).half(),
            torch.randint(-2**31, 2**31 - 1, (G, N // 8), dtype=torch.int32, device=dev, generator=gen), None)


def timed(pl, x, y):
    for _ in range(2): pl.forward(x, y)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(REPS): pl.forward(x, y)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / REPS


for K, N, ns in [(4096, 4096, 1), (4096, 12288, 1), (11008, 4096, 1), (4096, 11008, 2)]:
    sets = tuple(rand_set(K, N, K // 128) for _ in range(ns))
    pl = PreparedLayer(sets, None, 4, 128, K, N)
    for M in MS:
        x = (torch.randn((M, K), device=dev, generator=gen) * 0.5).half(); y = torch.empty((M, N), dtype=torch.float16, device=dev)
        row = {'shape': '%dx%d%s' % (K, N, ' pair' if ns == 2 else ''), 'M': M, 'two_copies_us': round(timed(pl, x, y), 1)}
        y0 = y.clone()
        pl2 = PreparedLayer(tuple(tuple(t.clone() if t is not None else None for t in s) for s in sets), None, 4, 128, K, N)
        pl2.release()
        row['released_us'] = round(timed(pl2, x, y), 1)
        row['identical'] = bool(torch.equal(y, y0))
        row['released_over_two_copies'] = round(row['released_us'] / row['two_copies_us'], 3)
        print(json.dumps(row), flush=True)
        del pl2
