#!/usr/bin/env python3
"""one prefill GEMM shape, a few launches (for rocprofv3 --pmc)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'gptq-for-llama_amd')); sys.path.insert(0, ROOT)
import torch
from bench import PackedSet, BITS, GS
from quant import quant_linear as QL
M = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
K, N = 4096, 4096
dev = 'cuda:0'
gen = torch.Generator(device=dev); gen.manual_seed(0)
w = PackedSet(K, N, dev, gen)
x = torch.randn((M, K), device=dev, generator=gen).half()
g_idx = (torch.arange(K, device=dev) // GS).to(torch.int32)
for _ in range(3):
    y = QL.matmul248(x, w.qweight, w.scales, w.qzeros, g_idx, BITS, 15)
torch.cuda.synchronize()
