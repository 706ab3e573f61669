#!/usr/bin/env python3
"""timing decomposition of the v3 prefill GEMM (development): which part of the loop costs what."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'gptq-for-llama_amd')); sys.path.insert(0, ROOT)
import torch
from bench import PackedSet, BITS, GS
from quant import _native, quant_linear as QL
M, K, N = 32768, 4096, 4096
dev = 'cuda:0'; lib = _native.lib()
gen = torch.Generator(device=dev); gen.manual_seed(0)
w = PackedSet(K, N, dev, gen); x = torch.randn((M, K), device=dev, generator=gen).half()
g_idx = (torch.arange(K, device=dev) // GS).to(torch.int32)
f = lambda: QL.matmul248(x, w.qweight, w.scales, w.qzeros, g_idx, BITS, 15, family='abi')
for name, v in [('full', 100), ('no MFMA', 101), ('no LDS reads / dequant', 102), ('LDS reads, no dequant', 103), ('no DMA in loop', 104), ('full again', 100)]:
    lib.gptq_set_gemm_kernel(v)
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): f()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    print('%-26s %.3f ms  (%.0f TFLOP/s-equivalent)' % (name, ms, 2.0 * M * N * K / ms / 1e9))
lib.gptq_set_gemm_kernel(100)
