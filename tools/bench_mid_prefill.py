#!/usr/bin/env python3
"""tools/bench_mid_prefill.py -- batches of 129 .. 4096 rows through gptq_layer_forward: the fused-dequantise tile GEMM on the stripe16
image (csrc/stripe_mm.inc, stripe_gemm_kernel) against the dense route (dequantise per call + gemm8 / hipBLASLt, whichever the prefill
switch picks) on the same prepared layer.  us per call (hipGraph of 8 back-to-back calls, median of 5 replays; weights warm: both routes read them once per call),
TFLOP/s = 2 M N K / t, and the max-normalised difference between the two results."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'gptq-for-llama_amd'))
sys.path.insert(0, ROOT)
import torch
from quant import _native, layer as QLayer

BITS, GS = int(os.environ.get('BITS', '4')), 128
dev = torch.device('cuda:0')
lib = _native.lib()
gen = torch.Generator(device=dev)
gen.manual_seed(5)
DEFAULT_ROWS = lib.gptq_set_stripe_gemm_max_rows(1024)
lib.gptq_set_stripe_gemm_max_rows(DEFAULT_ROWS)
MS = [int(m) for m in os.environ.get('MS', '129,256,512,1024,1536,2048,4096').split(',')]


def rand_set(K, N):
    G = K // GS
    qw = torch.randint(-2**31, 2**31 - 1, (K * BITS // 32, N), dtype=torch.int32, device=dev, generator=gen)
    qz = torch.randint(-2**31, 2**31 - 1, (G, N * BITS // 32), dtype=torch.int32, device=dev, generator=gen)
    sc = (torch.rand((G, N), device=dev, generator=gen) * 0.01 + 0.001).half()
    return qw, sc, qz, None


def timeit(fn, calls=8, reps=5):
    """us per call from a hipGraph of `calls` back-to-back calls (a single call is shorter than the Python + ctypes time to enqueue it)"""
    fn(); fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(calls):
            fn()
    g.replay()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); g.replay(); e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3 / calls)
    ts.sort()
    return ts[len(ts) // 2]


for (K, N, pair) in [(4096, 4096, False), (4096, 12288, False), (4096, 11008, False), (11008, 4096, False), (4096, 11008, True)]:
    sets = (rand_set(K, N), rand_set(K, N)) if pair else (rand_set(K, N),)
    pl = QLayer.PreparedLayer(sets, None, BITS, GS, K, N)
    for M in MS:
        x = torch.randn((M, K), device=dev, generator=gen).half()
        out = torch.empty((M, N), dtype=torch.float16, device=dev)
        res = {}
        # dense route = dequantise per call + a dense GEMM: `library` hipBLASLt (gptq_set_prefill_route(0): the reported ceiling), `gemm8` the
        # hand-written tile GEMM forced for every M (route 2), `auto` what the default switch picks; `image` = the fused tile GEMM on the stripe16 image
        for name, rows, route in (('library', 0, 0), ('gemm8', 0, 2), ('dense', 0, 1), ('image', 1 << 20, 1)):
            if name in ('library', 'gemm8') and not os.environ.get('ALL_ROUTES'):
                continue
            lib.gptq_set_stripe_gemm_max_rows(rows)
            lib.gptq_set_prefill_route(route)
            t = timeit(lambda: pl.forward(x, out))
            res[name] = (t, out.float().clone())
        lib.gptq_set_stripe_gemm_max_rows(DEFAULT_ROWS)
        lib.gptq_set_prefill_route(1)
        fl = (4.0 if pair else 2.0) * M * N * K
        d = float((res['dense'][1] - res['image'][1]).abs().max() / res['dense'][1].abs().max())
        extra = ''
        if 'library' in res:
            extra = ' | library %8.1f us %7.1f TF | gemm8 forced %8.1f us %7.1f TF' % (res['library'][0], fl / res['library'][0] / 1e6, res['gemm8'][0], fl / res['gemm8'][0] / 1e6)
        print('%s %5dx%-5d M=%-5d dense route %8.1f us %7.1f TF | tile GEMM on the image %8.1f us %7.1f TF | image / dense %.2fx | max rel diff %.1e%s'
              % ('pair' if pair else '    ', K, N, M, res['dense'][0], fl / res['dense'][0] / 1e6, res['image'][0], fl / res['image'][0] / 1e6,
                 res['dense'][0] / res['image'][0], d, extra), flush=True)
    del pl, sets
    torch.cuda.empty_cache()
