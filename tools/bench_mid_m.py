#!/usr/bin/env python3
"""the Python-level dispatch (quant_linear.matmul248) across M on LLaMA-7B shapes, cold-ish weights
(8 sets rotated): rowwave / stream kernel / dequantise + dense GEMM / MFMA tile GEMM."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'gptq-for-llama_amd')); sys.path.insert(0, ROOT)
import torch
from bench import PackedSet, BITS, GS
from quant import _native, quant_linear as QL
dev = 'cuda:0'
gen = torch.Generator(device=dev); gen.manual_seed(0)
for K, N in [(4096, 4096), (4096, 11008)]:
    sets = [PackedSet(K, N, dev, gen) for _ in range(8)]
    g_idx = (torch.arange(K, device=dev) // GS).to(torch.int32)
    for M in [int(v) for v in os.environ.get('MS', '1,8,16,32,64,96,128,192,256,512,1024,2048,4096,8192').split(',')]:
        x = torch.randn((M, K), device=dev, generator=gen).half()
        def run(i, fam=None):
            w = sets[i % 8]
            return QL.matmul248(x, w.qweight, w.scales, w.qzeros, g_idx, BITS, 15, family=fam)
        res = {}
        for fam in (None, 'abi') + (('stripe_mm',) if M <= 256 else ()):
            for i in range(8): run(i, fam)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for i in range(16): run(i, fam)
            e1.record(); torch.cuda.synchronize()
            res['dispatch' if fam is None else ('abi_only' if fam == 'abi' else 'stripe_mm')] = round(e0.elapsed_time(e1) * 1e3 / 16, 1)
        route = 'stripe16' if M <= 128 else ('tile GEMM (gemm8)' if _native.lib().gptq_prefill_route_for(M, K, N, 1, 0) == 1 else 'dequant + hipBLASLt')
        print(json.dumps({'shape': '%dx%d' % (K, N), 'M': M, 'route': route, 'us': res['dispatch'], 'us_abi_kernels_only': res['abi_only'], 'us_stripe_mm': res.get('stripe_mm'),
                          'TFLOPs': round(2.0 * M * K * N / res['dispatch'] / 1e6, 1)}))
