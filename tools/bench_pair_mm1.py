#!/usr/bin/env python3
"""the single-launch gate/up pair kernel for 5..16 rows (stripe_mm1_kernel<1, 2, PF>): PF = 2 (two workgroups per CU, 128 VGPRs, 208 B of
scratch per lane) against PF = 4 (one workgroup per CU, no spills).  Run once per setting: GPTQ_MM1_PAIR_PF=2|4 python tools/bench_pair_mm1.py
Round 6: MS=16,32,..,128 times 2 x 4096 x 11008 at those batches (17 .. 128 rows: the loader / consumer pair, GPTQ_MMR_PAIR=0 restores the routes before)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'gptq-for-llama_amd')); sys.path.insert(0, ROOT)
import torch
from bench import PackedSet, alg_bytes, BITS, GS
from quant import _native, quant_linear as QL
dev = 'cuda:0'; lib = _native.lib()
gen = torch.Generator(device=dev); gen.manual_seed(0)
mws = _native.mm_workspace(torch.device(dev))
for K, N in ([(4096, 11008)] if os.environ.get('MS') else [(4096, 11008), (4096, 4096)]):
    nsets = int(300e6 // alg_bytes(1, K, N, nsets=2)) + 1
    imgs = []
    for _ in range(nsets):
        a, b = PackedSet(K, N, dev, gen), PackedSet(K, N, dev, gen)
        imgs.append(QL.stripe_copy(a.qweight, a.scales, a.qzeros, BITS, GS, up=(b.qweight, b.scales, b.qzeros)))
        torch.cuda.synchronize()
        del a, b
    for M in [int(v) for v in os.environ.get('MS', '9,16').split(',')]:
        x = torch.randn((M, K), device=dev, generator=gen).half(); y = torch.empty((M, N), dtype=torch.float16, device=dev)

        def run(i):
            st = imgs[i]
            rc = lib.gptq_stripe_matmul_f16(x.data_ptr(), K, st.data_ptr(), st.numel(), None, y.data_ptr(), N, M, K, N, BITS, GS, 2, mws.data_ptr(), mws.numel(),
                                            torch.cuda.current_stream().cuda_stream)
            _native.check(rc, 'stripe_matmul pair')
        for i in range(nsets): run(i)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for i in range(nsets): run(i)
        g.replay(); torch.cuda.synchronize()
        best = 1e9
        for _ in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5): g.replay()
            e1.record(); torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) * 1e3 / (5 * nsets))
        print('MMR_PAIR=%s PF=%s gate/up 2x%dx%d M=%-2d: %.2f us' % (os.environ.get('GPTQ_MMR_PAIR', 'default(1)'), os.environ.get('GPTQ_MM1_PAIR_PF', 'default(4)'), K, N, M, best), flush=True)
    del imgs
    torch.cuda.empty_cache()
