#!/usr/bin/env python3
"""M = 1 .. 256 on the 7B shapes through gptq_matmul248_f16 (cold weights, hipGraph): which kernel
family serves each M and how far from the HBM / MFMA roofline it sits."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'gptq-for-llama_amd')); sys.path.insert(0, ROOT)
import torch
from bench import PackedSet, alg_bytes, BITS, GS
from quant import _native
dev = 'cuda:0'; lib = _native.lib(); ws = _native.workspace(torch.device(dev))
gen = torch.Generator(device=dev); gen.manual_seed(0)
if os.environ.get('GEMV_VARIANT'):
    lib.gptq_set_gemv_variant(int(os.environ['GEMV_VARIANT']))   # 100 = dot2 small-batch kernel instead of the MFMA 4x4x4 one
if os.environ.get('SPLIT_K'):
    lib.gptq_set_split_k(int(os.environ['SPLIT_K']))
MS = [int(v) for v in os.environ.get('MS', '1,2,3,4,5,8,16,32,64,128,256').split(',')]
for K, N in [(4096, 4096), (4096, 11008)]:
    nsets = int(300e6 // alg_bytes(1, K, N)) + 1
    sets = [PackedSet(K, N, dev, gen) for _ in range(nsets)]
    for M in MS:
        x = torch.randn((M, K), device=dev, generator=gen).half(); y = torch.empty((M, N), dtype=torch.float16, device=dev)
        def run(i):
            w = sets[i]
            rc = lib.gptq_matmul248_f16(x.data_ptr(), K, w.qweight.data_ptr(), w.scales.data_ptr(), w.qzeros.data_ptr(), None, None,
                                        y.data_ptr(), N, M, K, N, BITS, GS, ws.data_ptr(), ws.numel(), torch.cuda.current_stream().cuda_stream)
            _native.check(rc, 'mm')
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
        nb = alg_bytes(M, K, N)
        print(json.dumps({'shape': '%dx%d' % (K, N), 'M': M, 'us': round(us, 2), 'GBps': round(nb / us / 1e3, 1), 'TFLOPs': round(2.0 * M * K * N / us / 1e6, 2)}))
        del g
