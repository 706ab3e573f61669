#!/usr/bin/env python3
"""phase stamps of the prefill GEMM's ping-pong loop (slab 8 of the first 64 workgroups)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'gptq-for-llama_amd')); sys.path.insert(0, ROOT)
import numpy as np, torch
from bench import PackedSet, BITS, GS
from quant import _native, quant_linear as QL
M, K, N = 16384, 4096, 4096
dev = 'cuda:0'; lib = _native.lib()
gen = torch.Generator(device=dev); gen.manual_seed(0)
w = PackedSet(K, N, dev, gen)
x = torch.randn((M, K), device=dev, generator=gen).half()
g_idx = (torch.arange(K, device=dev) // GS).to(torch.int32)
f = lambda: QL.matmul248(x, w.qweight, w.scales, w.qzeros, g_idx, BITS, 15)
f(); torch.cuda.synchronize()
dbg = torch.zeros(64 * 8 * 8, dtype=torch.int64, device=dev)
lib.gptq_set_debug_buffer(dbg.data_ptr()); f(); torch.cuda.synchronize(); lib.gptq_set_debug_buffer(None)
d = dbg.cpu().numpy().reshape(64, 8, 8)
for name, a, b in [('compute', 0, 1), ('barrier after compute', 1, 2), ('move: wait + A ds_writes', 2, 5), ('move: dequant + B ds_writes', 5, 6), ('move: issue next loads', 6, 3), ('move', 2, 3), ('barrier after move', 3, 4), ('whole iteration', 0, 4)]:
    for st, sl in (('set0', slice(0, 4)), ('set1', slice(4, 8))):
        v = (d[:, sl, b] - d[:, sl, a]).astype(np.float64).ravel()
        print('%-22s %s  p10 %7.0f  p50 %7.0f  p90 %7.0f cycles' % (name, st, np.percentile(v, 10), np.median(v), np.percentile(v, 90)))
