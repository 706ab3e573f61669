#!/usr/bin/env python3
"""tools/bench_lm_head.py -- the LM head of a decode step (dense fp16 32000 x 4096 matvec, 262 MB): gptq_dense_matvec_f16 (csrc/dense_gemv.hip,
with and without the fused final RMSNorm) against RMSNorm launch + torch.matmul (hipBLASLt), cold weights (rotation over copies > 256 MiB),
us per call from one hipGraph."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'gptq-for-llama_amd'))
import torch
from quant import _native

dev = torch.device('cuda:0')
lib = _native.lib()
N, K = int(os.environ.get('N', '32000')), int(os.environ.get('K', '4096'))
copies = max(2, int(600e6 // (N * K * 2)) + 1)
Ws = [(torch.randn((N, K), device=dev) * 0.02).half() for _ in range(copies)]
x = torch.randn((1, K), device=dev).half()
nw = (1 + 0.1 * torch.randn(K, device=dev)).half()
h = torch.empty_like(x)
y = torch.empty((1, N), dtype=torch.float16, device=dev)


def own(W, norm):
    rc = lib.gptq_dense_matvec_f16(x.data_ptr(), W.data_ptr(), K, None, y.data_ptr(), N, K, nw.data_ptr() if norm else None, 1e-6,
                                   torch.cuda.current_stream().cuda_stream)
    _native.check(rc, 'gptq_dense_matvec_f16')


def library(W):
    rc = lib.gptq_rmsnorm_f16(x.data_ptr(), K, nw.data_ptr(), h.data_ptr(), K, 1, K, 1e-6, torch.cuda.current_stream().cuda_stream)
    _native.check(rc, 'gptq_rmsnorm_f16')
    torch.matmul(h, W.t(), out=y)


def timeit(fn, reps=10):
    for W in Ws:
        fn(W)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(4):
            for W in Ws:
                fn(W)
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (reps * 4 * len(Ws))


nbytes = N * K * 2 + K * 2 + N * 2
for name, fn in (('own kernel', lambda W: own(W, False)), ('own kernel + fused final RMSNorm', lambda W: own(W, True)),
                 ('RMSNorm launch + torch.matmul (hipBLASLt)', library)):
    us = timeit(fn)
    print('%-45s %7.2f us  %6.0f GB/s  (%.3f of 8 TB/s)' % (name, us, nbytes / us / 1e3, nbytes / us / 1e3 / 8000), flush=True)
own(Ws[0], True)
a = y.float().clone()
library(Ws[0])
print('max rel diff own vs library: %.2e' % float((a - y.float()).abs().max() / y.float().abs().max()))
