#!/usr/bin/env python3
"""tools/act_order_probe.py -- what does act-order cost at M = 1?  The same 4-bit g128 layer shape with a trivial g_idx, with a regular act-order
g_idx whose permutation is the identity except for two swapped rows, and with a random one (cold weights, hipGraph): the gather PATTERN is
irrelevant (swap == random); profiles/r3i_act_order/.  BITS=3 (round 4: 3-bit act-order layers get a group-sorted image too); SORT=0 = the generic
g_idx kernels on the checkpoint layout, what such a layer ran on before."""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'gptq-for-llama_amd')); sys.path.insert(0, ROOT)
import torch
from quant import _native, quant_linear as QL
dev = 'cuda:0'
BITS = int(os.environ.get('BITS', '4'))
QL.ACT_ORDER_SORT = os.environ.get('SORT', '1') != '0'
gen = torch.Generator(device=dev); gen.manual_seed(0)
def make(K, N, mode):
    G = K // 128
    qw = torch.randint(-2**31, 2**31 - 1, (K // 32 * BITS, N), dtype=torch.int32, device=dev, generator=gen)
    qz = torch.randint(-2**31, 2**31 - 1, (G, N // 32 * BITS), dtype=torch.int32, device=dev, generator=gen)
    sc = (torch.rand((G, N), device=dev, generator=gen) * 0.01 + 0.001).half()
    gi = (torch.arange(K, device=dev) // 128).to(torch.int32)
    if mode == 'swap':       # regular act-order whose permutation is the identity except for two rows of different groups
        gi = gi.clone(); gi[0], gi[K - 1] = gi[K - 1].clone(), gi[0].clone()
    elif mode == 'random':
        gi = gi[torch.argsort(torch.randperm(K, device=dev, generator=gen))].contiguous()
    return qw, sc, qz, gi
for K, N in [(4096, 4096), (11008, 4096)]:
    for mode in ['trivial', 'swap', 'random']:
        nb = K * N * BITS // 8
        nsets = int(300e6 // nb) + 1
        sets = [make(K, N, mode) for _ in range(nsets)]
        x = torch.randn((1, K), device=dev, generator=gen).half()
        def run(i):
            qw, sc, qz, gi = sets[i]
            return QL.matmul248(x, qw, sc, qz, gi, BITS, 2**BITS - 1)
        for i in range(nsets): run(i)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for i in range(nsets): run(i)
        g.replay(); torch.cuda.synchronize()
        best = 1e9
        for _ in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5): g.replay()
            e1.record(); torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) * 1e3 / (5 * nsets))
        print('w%d %dx%d %-8s %.2f us' % (BITS, K, N, mode, best), flush=True)
        del sets, g
        torch.cuda.empty_cache()
