#!/usr/bin/env python3
"""Backward product dx = dy . deq(W)^T (reference transpose_matmul248, quant_linear.py:272-279): the LDS-tiled kernel of
csrc/transpose.hip (family='abi') against the prefill route (dequantise per call + hipBLASLt with the transposition flag),
LLaMA-7B shapes, 4-bit g128.  Decides TRANSPOSE_LIBRARY_MIN_M.
usage: python tools/bench_backward.py [--ms 1,8,16,64,512,4096]"""
import argparse, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'gptq-for-llama_amd')); sys.path.insert(0, ROOT)
import torch
from bench import PackedSet, BITS, GS
from quant import quant_linear as QL

ap = argparse.ArgumentParser()
ap.add_argument('--ms', default='1,8,16,64,512,4096')
a = ap.parse_args()
dev = 'cuda:0'
gen = torch.Generator(device=dev); gen.manual_seed(0)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)


def timed(f, reps):
    f(); f(); torch.cuda.synchronize()
    e0.record()
    for _ in range(reps):
        f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


for K, N in [(4096, 4096), (4096, 11008)]:
    w = PackedSet(K, N, dev, gen)
    gi = (torch.arange(K, device=dev) // GS).to(torch.int32)
    for M in [int(v) for v in a.ms.split(',')]:
        dy = torch.randn((M, N), device=dev, generator=gen).half()
        QL.TRANSPOSE_LIBRARY_MIN_M = 1
        lib = lambda: QL.transpose_matmul248(dy, w.qweight, w.scales, w.qzeros, gi, BITS, 15)
        own = lambda: QL.transpose_matmul248(dy, w.qweight, w.scales, w.qzeros, gi, BITS, 15, family='abi')
        reps = 3 if M >= 4096 else 10
        t_l, t_o = timed(lib, reps), timed(own, reps)
        fl = 2.0 * M * N * K / 1e9
        print(json.dumps({'shape': '%dx%d' % (K, N), 'M': M, 'own_kernel_ms': round(t_o, 4), 'library_route_ms': round(t_l, 4),
                          'own_TF': round(fl / t_o, 2), 'library_TF': round(fl / t_l, 2),
                          'max_abs_diff': float((lib().float() - own().float()).abs().max())}), flush=True)
