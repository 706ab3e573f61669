#!/usr/bin/env python3
"""a few launches of gptq_stripe_matmul_f16 at ONE (shape, M) on cold weights -- for rocprofv3 --kernel-trace / --pmc runs.
usage: run_small_batch_once.py K N M [reps]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'gptq-for-llama_amd')); sys.path.insert(0, ROOT)
import torch
from quant import _native, quant_linear as QL
K, N, M = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 3
dev = 'cuda:0'; lib = _native.lib(); gen = torch.Generator(device=dev); gen.manual_seed(0)
nsets = int(300e6 // (K * N // 2)) + 1
imgs = []
for _ in range(nsets):
    qw = torch.randint(-2**31, 2**31 - 1, (K // 8, N), dtype=torch.int32, device=dev, generator=gen)
    qz = torch.randint(-2**31, 2**31 - 1, (K // 128, N // 8), dtype=torch.int32, device=dev, generator=gen)
    sc = (torch.rand((K // 128, N), device=dev, generator=gen) * 0.01 + 0.001).half()
    imgs.append(QL.stripe_copy(qw, sc, qz, 4, 128))
x = torch.randn((M, K), device=dev, generator=gen).half(); y = torch.empty((M, N), dtype=torch.float16, device=dev)
mws = _native.mm_workspace(torch.device(dev))
for _ in range(reps):
    for st in imgs:
        _native.check(lib.gptq_stripe_matmul_f16(x.data_ptr(), K, st.data_ptr(), st.numel(), None, y.data_ptr(), N, M, K, N, 4, 128, 1, mws.data_ptr(), mws.numel(),
                                                 torch.cuda.current_stream().cuda_stream), 'mm')
torch.cuda.synchronize()
print('done', K, N, M, nsets * reps, 'launches')
