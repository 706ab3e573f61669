#!/usr/bin/env python3
"""BASELINE config 4: LLaMA-7B-shaped 3-bit no-group + 4-bit act-order (g128), batch 1, cold weights
(rotating over > 256 MiB of distinct sets inside one hipGraph).  GB/s uses the SURVEY 8(d) byte model
(act-order adds 4*K bytes of g_idx)."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'gptq-for-llama_amd')); sys.path.insert(0, ROOT)
import torch
from quant import _native, quant_linear as QL
dev = 'cuda:0'
lib = _native.lib(); ws = _native.workspace(torch.device(dev))
gen = torch.Generator(device=dev); gen.manual_seed(0)

def make(bits, gs, K, N, act):
    G = 1 if gs == -1 else K // gs
    qw = torch.randint(-2**31, 2**31 - 1, (K // 32 * bits, N), dtype=torch.int32, device=dev, generator=gen)
    qz = torch.randint(-2**31, 2**31 - 1, (G, N // 32 * bits), dtype=torch.int32, device=dev, generator=gen)
    sc = (torch.rand((G, N), device=dev, generator=gen) * 0.01 + 0.001).half()
    gsz = K if gs == -1 else gs
    gi = (torch.arange(K, device=dev) // gsz).to(torch.int32)
    if act:
        gi = gi[torch.argsort(torch.randperm(K, device=dev, generator=gen))].contiguous()
    return qw, sc, qz, gi

def bytes_model(bits, gs, K, N, act):
    G = 1 if gs == -1 else K // gs
    return 4 * (K * bits // 32) * N + 4 * G * (N * bits // 32) + 2 * G * N + 2 * K + 2 * N + (4 * K if act else 0)

for label, bits, gs, act in [('3-bit no-group', 3, -1, False), ('4-bit g128 act-order', 4, 128, True), ('4-bit g128 (trivial g_idx, for scale)', 4, 128, False)]:
    for K, N in [(4096, 4096), (4096, 12288), (11008, 4096), (4096, 11008)]:
        nb = bytes_model(bits, gs, K, N, act)
        nsets = int(300e6 // nb) + 1
        sets = [make(bits, gs, K, N, act) for _ in range(nsets)]
        x = torch.randn((1, K), device=dev, generator=gen).half()
        prep = getattr(QL, 'prepare_act_order', None)
        def run(i):
            qw, sc, qz, gi = sets[i]
            return QL.matmul248(x, qw, sc, qz, gi, bits, 2**bits - 1)
        for i in range(nsets): run(i)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for i in range(nsets): run(i)
        g.replay(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5): g.replay()
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / (5 * nsets)
        print(json.dumps({'config': label, 'shape': '%dx%d' % (K, N), 'us': round(us, 2), 'GBps': round(nb / us / 1e3, 1), 'frac_of_8TBps': round(nb / us / 8e6, 4)}))
        del sets, g
        torch.cuda.empty_cache()

# fused gate/up + SiLU of an act-order MLP (gate and up share the permutation): group-sorted fast path vs the generic g_idx kernel
from quant import fused_mlp as FM
K, N = 4096, 11008
nb = 2 * (bytes_model(4, 128, K, N, True) - 2 * K - 2 * N) + 2 * K + 2 * N
nsets = int(300e6 // nb) + 1
gi = (torch.arange(K, device=dev) // 128).to(torch.int32)[torch.argsort(torch.randperm(K, device=dev, generator=gen))].contiguous()
sets = [(make(4, 128, K, N, False), make(4, 128, K, N, False)) for _ in range(nsets)]
x = torch.randn((1, K), device=dev, generator=gen).half()
for label, flag in [('4-bit g128 act-order fused gate/up, sorted fast path', True), ('4-bit g128 act-order fused gate/up, generic kernel', False)]:
    QL.ACT_ORDER_SORT = flag
    def run(i):
        (a, b) = sets[i]
        return FM.fused_gate_up(x, (a[0], a[1], a[2], gi), (b[0], b[1], b[2], gi), 4, 128)
    for i in range(nsets): run(i)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for i in range(nsets): run(i)
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): g.replay()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / (5 * nsets)
    print(json.dumps({'config': label, 'shape': '2x%dx%d' % (K, N), 'us': round(us, 2), 'GBps': round(nb / us / 1e3, 1), 'frac_of_8TBps': round(nb / us / 8e6, 4)}))
    del g
QL.ACT_ORDER_SORT = True

# 3-bit no-group fused gate/up + SiLU (gemv_rowwave3_kernel<UB, true>)
K, N = 4096, 11008
nb = 2 * (bytes_model(3, -1, K, N, False) - 2 * K - 2 * N) + 2 * K + 2 * N
nsets = int(300e6 // nb) + 1
sets = [(make(3, -1, K, N, False), make(3, -1, K, N, False)) for _ in range(nsets)]
def run(i):
    (a, b) = sets[i]
    return FM.fused_gate_up(x, (a[0], a[1], a[2], a[3]), (b[0], b[1], b[2], b[3]), 3, K)
for i in range(nsets): run(i)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    for i in range(nsets): run(i)
g.replay(); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(5): g.replay()
e1.record(); torch.cuda.synchronize()
us = e0.elapsed_time(e1) * 1e3 / (5 * nsets)
print(json.dumps({'config': '3-bit no-group fused gate/up', 'shape': '2x%dx%d' % (K, N), 'us': round(us, 2), 'GBps': round(nb / us / 1e3, 1), 'frac_of_8TBps': round(nb / us / 8e6, 4)}))
