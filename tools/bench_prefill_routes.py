#!/usr/bin/env python3
"""Prefill routes side by side (BASELINE config 3 shapes, 4-bit g128): the fused MFMA tile GEMM (csrc/gemm_mfma.hip, through
family='abi') against "dequantise once (gptq_dequant_f16) + library GEMM" INCLUDING the dequantise pass and its transient
fp16 weight, per call -- the number that decides where matmul248 hands a batch to which route.
usage: python tools/bench_prefill_routes.py [--ms 256,1024,4096,16384,65536] [--reps 5]"""
import argparse, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'gptq-for-llama_amd')); sys.path.insert(0, ROOT)
import torch
from bench import PackedSet, BITS, GS
from quant import quant_linear as QL

ap = argparse.ArgumentParser()
ap.add_argument('--ms', default='256,1024,4096,16384,65536'); ap.add_argument('--reps', type=int, default=5)
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


for K, N in [(4096, 4096), (4096, 12288), (4096, 11008), (11008, 4096)]:
    w = PackedSet(K, N, dev, gen)
    gi = (torch.arange(K, device=dev) // GS).to(torch.int32)
    for M in [int(v) for v in a.ms.split(',')]:
        x = torch.randn((M, K), device=dev, generator=gen).half()
        out = torch.empty((M, N), device=dev, dtype=torch.float16)

        def lib():
            W = QL.dequantize(w.qweight, w.scales, w.qzeros, None, BITS, GS)
            torch.matmul(x, W, out=out)

        fused = lambda: QL.matmul248(x, w.qweight, w.scales, w.qzeros, gi, BITS, 15, family='abi')
        deq = lambda: QL.dequantize(w.qweight, w.scales, w.qzeros, None, BITS, GS)
        reps = a.reps if M >= 16384 else 4 * a.reps
        prod = lambda: QL.matmul248(x, w.qweight, w.scales, w.qzeros, gi, BITS, 15)       # the built-in dispatch (GPTQ_PREFILL)
        t_f, t_l, t_d, t_p = timed(fused, reps), timed(lib, reps), timed(deq, reps), timed(prod, reps)
        yf = fused().float(); lib()
        fl = 2.0 * M * N * K / 1e9
        print(json.dumps({'shape': '%dx%d' % (K, N), 'M': M, 'fused_ms': round(t_f, 4), 'fused_TF': round(fl / t_f, 1),
                          'dequant_plus_library_ms': round(t_l, 4), 'dequant_plus_library_TF': round(fl / t_l, 1),
                          'dequant_ms': round(t_d, 4), 'matmul248_default_ms': round(t_p, 4), 'matmul248_default_TF': round(fl / t_p, 1), 'fused_over_library': round(t_l / t_f, 3),
                          'max_abs_diff': float((yf - out.float()).abs().max())}), flush=True)
        del x, out, yf
        torch.cuda.empty_cache()
