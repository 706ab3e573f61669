#!/usr/bin/env python3
"""stripe16 small-batch MFMA kernel (csrc/stripe_mm.inc) vs the 4x4x4 row groups of the decode kernel and the rowwave / skinny
kernels of gptq_matmul248_f16: us per launch on cold weights (hipGraph over >= 300 MB of weight sets), plus a torch fp32 check."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'gptq-for-llama_amd')); sys.path.insert(0, ROOT)
import torch
from bench import alg_bytes, GS
from quant import _native, quant_linear as QL
dev = 'cuda:0'; lib = _native.lib(); ws = _native.workspace(torch.device(dev))
gen = torch.Generator(device=dev); gen.manual_seed(0)
if os.environ.get('PASS_ROWS'):      # 64 = round 2's schedule (two passes for 65..128 rows), 128 = one pass with six / eight row tiles
    lib.gptq_set_stripe_mm_pass_rows(int(os.environ['PASS_ROWS']))
MS = [int(v) for v in os.environ.get('MS', '1,4,5,8,16,17,32,48,64').split(',')]
BITS = int(os.environ.get('BITS', '4'))
SKS = [-1] + [int(v) for v in os.environ.get('SKS', '').split(',') if v]
SHAPES = [(4096, 4096), (4096, 12288), (11008, 4096)]
if os.environ.get('SHAPES'):
    SHAPES = [tuple(int(v) for v in s.split('x')) for s in os.environ['SHAPES'].split(',')]


class PackedSet:
    def __init__(self, K, N, dev, gen, bits):
        G = K // GS
        self.qweight = torch.randint(-2**31, 2**31 - 1, (K * bits // 32, N), dtype=torch.int32, device=dev, generator=gen)
        self.qzeros = torch.randint(-2**31, 2**31 - 1, (G, N * bits // 32), dtype=torch.int32, device=dev, generator=gen)
        self.scales = (torch.rand((G, N), device=dev, generator=gen) * 0.01 + 0.001).half()


def timed(fn, nsets):
    for i in range(nsets): fn(i)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for i in range(nsets): fn(i)
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): g.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (5 * nsets)


for K, N in SHAPES:
    nsets = int(300e6 // alg_bytes(1, K, N, bits=BITS)) + 1
    sets = [PackedSet(K, N, dev, gen, BITS) for _ in range(nsets)]
    imgs = [QL.stripe_copy(w.qweight, w.scales, w.qzeros, BITS, GS) for w in sets]
    w0 = sets[0]
    Wd = QL.dequantize(w0.qweight, w0.scales, w0.qzeros, None, BITS, GS).float()
    for M in MS:
        x = torch.randn((M, K), device=dev, generator=gen).half(); y = torch.empty((M, N), dtype=torch.float16, device=dev)
        sp = torch.cuda.current_stream().cuda_stream

        def run_stripe(i):
            st = imgs[i]
            rc = lib.gptq_stripe_matvec_f16(x.data_ptr(), K, st.data_ptr(), st.numel() * st.element_size(), None, y.data_ptr(), N, M, K, N, BITS, GS, 1, None, 0.0, None,
                                            torch.cuda.current_stream().cuda_stream)
            _native.check(rc, 'stripe')

        def run_mm(i):
            st = imgs[i]
            mws = _native.mm_workspace(torch.device(dev))
            rc = lib.gptq_stripe_matmul_f16(x.data_ptr(), K, st.data_ptr(), st.numel() * st.element_size(), None, y.data_ptr(), N, M, K, N, BITS, GS, 1,
                                            mws.data_ptr(), mws.numel(), torch.cuda.current_stream().cuda_stream)
            _native.check(rc, 'stripe_mm')

        def run_abi(i):
            w = sets[i]
            rc = lib.gptq_matmul248_f16(x.data_ptr(), K, w.qweight.data_ptr(), w.scales.data_ptr(), w.qzeros.data_ptr(), None, None,
                                        y.data_ptr(), N, M, K, N, BITS, GS, ws.data_ptr(), ws.numel(), torch.cuda.current_stream().cuda_stream)
            _native.check(rc, 'mm')
        row = {'bits': BITS, 'shape': '%dx%d' % (K, N), 'M': M}
        try:
            for sk in SKS:
                lib.gptq_set_split_k(sk)
                try:
                    row['mfma_us' if sk < 0 else 'mfma_S%d_us' % sk] = round(timed(run_mm, nsets), 2)
                except RuntimeError:
                    torch.cuda.synchronize()
                    continue
                y.zero_(); run_mm(0); torch.cuda.synchronize()
                ref = x.float() @ Wd
                err = float((y.float() - ref).abs().max() / ref.abs().max())
                if err > 1e-3:
                    row['BAD_S%d' % sk] = err
            lib.gptq_set_split_k(-1)
            y.zero_(); run_mm(0); torch.cuda.synchronize()
            if Wd is not None:
                ref = x.float() @ Wd
                row['mfma_relerr'] = float((y.float() - ref).abs().max() / ref.abs().max())
            row['mfma_TFLOPs'] = round(2.0 * M * K * N / row['mfma_us'] / 1e6, 1)
        except RuntimeError as e:
            row['mfma_us'] = None; row['err'] = str(e)[:80]
        lib.gptq_set_split_k(-1)
        if M <= 16:
            try:
                row['rowgroup_us'] = round(timed(run_stripe, nsets), 2)
            except RuntimeError:
                row['rowgroup_us'] = None
        row['abi_us'] = round(timed(run_abi, nsets), 2)
        print(json.dumps(row), flush=True)
    del sets, imgs
