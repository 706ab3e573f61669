#!/usr/bin/env python3
"""tools/bench_gemm8.py -- the hand-written prefill GEMM (csrc/gemm8.hip, route 1 of gptq_prefill_*) against hipBLASLt on the same
dequantised weight (route 0): correctness on awkward shapes (ragged M, N tails, bias, the gate/up pair, the backward product),
a run-to-run race screen (the kernel is deterministic: any differing bit between repeats is a synchronisation bug), and TFLOP/s
at the BASELINE config 3 sizes.  Random data everywhere (zero-filled operands clock higher: MI355X guide 5.4 rule 25)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'gptq-for-llama_amd'))
sys.path.insert(0, ROOT)

import torch
from quant import _native, quant_linear

BITS, GS = 4, 128
dev = torch.device('cuda:0')
lib = _native.lib()
gen = torch.Generator(device=dev)
gen.manual_seed(1)


def rand_set(K, N, gs=GS):
    G = K // gs
    qw = torch.randint(-2**31, 2**31 - 1, (K * BITS // 32, N), dtype=torch.int32, device=dev, generator=gen)
    qz = torch.randint(-2**31, 2**31 - 1, (G, N * BITS // 32), dtype=torch.int32, device=dev, generator=gen)
    sc = (torch.rand((G, N), device=dev, generator=gen) * 0.01 + 0.001).half()
    return qw, sc, qz


def stream():
    return torch.cuda.current_stream().cuda_stream


def prefill(route, x, w, bias=None, up=None):
    qw, sc, qz = w
    M, K = x.shape
    N = qw.shape[1]
    lib.gptq_set_prefill_route(2 if route else 0)
    y = torch.empty((M, N), dtype=torch.float16, device=dev)
    ws = torch.empty(lib.gptq_prefill_workspace_bytes(M, K, N, 2 if up else 1), dtype=torch.uint8, device=dev)
    if up is None:
        rc = lib.gptq_prefill_matmul_f16(x.data_ptr(), x.stride(0), qw.data_ptr(), sc.data_ptr(), qz.data_ptr(), None, _native.ptr(bias),
                                         y.data_ptr(), N, M, K, N, BITS, GS, ws.data_ptr(), ws.numel(), stream())
    else:
        rc = lib.gptq_prefill_fused_mlp_f16(x.data_ptr(), x.stride(0), qw.data_ptr(), sc.data_ptr(), qz.data_ptr(), None, up[0].data_ptr(),
                                            up[1].data_ptr(), up[2].data_ptr(), None, y.data_ptr(), N, M, K, N, BITS, GS, ws.data_ptr(), ws.numel(),
                                            stream())
    _native.check(rc, 'prefill route %d' % route)
    return y


def backward(route, dy, w):
    qw, sc, qz = w
    M, N = dy.shape
    K = qw.shape[0] * 32 // BITS
    lib.gptq_set_prefill_route(2 if route else 0)
    dx = torch.empty((M, K), dtype=torch.float16, device=dev)
    ws = torch.empty(lib.gptq_prefill_workspace_bytes(M, K, N, 1), dtype=torch.uint8, device=dev)
    rc = lib.gptq_prefill_transpose_matmul248_f16(dy.data_ptr(), dy.stride(0), qw.data_ptr(), sc.data_ptr(), qz.data_ptr(), None, dx.data_ptr(), K,
                                                  M, K, N, BITS, GS, ws.data_ptr(), ws.numel(), stream())
    _native.check(rc, 'backward route %d' % route)
    return dx


def check():
    ok = True
    print('== correctness (own tile GEMM vs fp32 reference on the dequantised weight; hipBLASLt beside it) ==')
    for (M, K, N) in [(256, 128, 256), (300, 256, 512), (1000, 4096, 4096), (257, 1024, 11008), (4096, 11008, 4096), (65, 256, 96 * 32), (513, 384, 288)]:
        w = rand_set(K, N)
        x = torch.randn((M, K), device=dev, generator=gen).half()
        bias = (torch.randn(N, device=dev, generator=gen) * 0.1).half()
        W = quant_linear.dequantize(w[0], w[1], w[2], None, BITS, GS)
        ref = x.float() @ W.float()
        for b in (None, bias):
            r = ref if b is None else (ref.half().float() + b.float())
            ys = [prefill(1, x, w, bias=b) for _ in range(3)]
            yl = prefill(0, x, w, bias=b)
            torch.cuda.synchronize()
            e_own = float((ys[0].float() - r).abs().max() / r.abs().max())
            e_lib = float((yl.float() - r).abs().max() / r.abs().max())
            same = all(torch.equal(ys[0], y) for y in ys[1:])
            d = float((ys[0].float() - yl.float()).abs().max())
            good = e_own < 1e-3 and same
            ok &= good
            print('  matmul M=%-5d K=%-5d N=%-5d bias=%d: own rel err %.2e (library %.2e), own vs library max |diff| %.3g, repeats identical %s  %s'
                  % (M, K, N, b is not None, e_own, e_lib, d, same, 'ok' if good else 'FAILED'))
    for (M, K, N) in [(256, 256, 256), (700, 4096, 11008), (129, 512, 160)]:
        g, u = rand_set(K, N), rand_set(K, N)
        x = torch.randn((M, K), device=dev, generator=gen).half()
        Wg = quant_linear.dequantize(g[0], g[1], g[2], None, BITS, GS).double()
        Wu = quant_linear.dequantize(u[0], u[1], u[2], None, BITS, GS).double()
        a, b = x.double() @ Wg, x.double() @ Wu
        ref = a * torch.sigmoid(a) * b
        ys = [prefill(1, x, g, up=u) for _ in range(3)]
        yl = prefill(0, x, g, up=u)
        torch.cuda.synchronize()
        e_own = float((ys[0].double() - ref).abs().max() / ref.abs().max())
        e_lib = float((yl.double() - ref).abs().max() / ref.abs().max())
        same = all(torch.equal(ys[0], y) for y in ys[1:])
        good = e_own < 1e-3 and same
        ok &= good
        print('  gate/up M=%-5d K=%-5d N=%-5d: own rel err %.2e (library fp32-SiLU route %.2e), repeats identical %s  %s' % (M, K, N, e_own, e_lib, same,
                                                                                                                          'ok' if good else 'FAILED'))
    for (M, K, N) in [(256, 256, 256), (1000, 4096, 4096), (333, 4096, 11008), (512, 11008, 4096)]:
        w = rand_set(K, N)
        dy = torch.randn((M, N), device=dev, generator=gen).half()
        W = quant_linear.dequantize(w[0], w[1], w[2], None, BITS, GS)
        ref = dy.float() @ W.float().t()
        dxs = [backward(1, dy, w) for _ in range(3)]
        dl = backward(0, dy, w)
        torch.cuda.synchronize()
        e_own = float((dxs[0].float() - ref).abs().max() / ref.abs().max())
        e_lib = float((dl.float() - ref).abs().max() / ref.abs().max())
        same = all(torch.equal(dxs[0], d) for d in dxs[1:])
        good = e_own < 1e-3 and same
        ok &= good
        print('  backward M=%-5d K=%-5d N=%-5d: own rel err %.2e (library %.2e), repeats identical %s  %s' % (M, K, N, e_own, e_lib, same, 'ok' if good else 'FAILED'))
    return ok


def timeit(fn, reps):
    fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / reps)
    return best


def perf():
    print('== TFLOP/s (2 M N K / time; the per-call dequantise pass is INSIDE both timings) ==')
    shapes = [(4096, 4096), (4096, 12288), (4096, 11008), (11008, 4096)]
    for M in (int(m) for m in os.environ.get('MS', '65536,16384,4096,2048,1024,256').split(',')):
        for (K, N) in shapes:
            w = rand_set(K, N)
            x = torch.randn((M, K), device=dev, generator=gen).half()
            reps = 3 if M >= 16384 else 10
            t1 = timeit(lambda: prefill(1, x, w), reps)
            t0 = timeit(lambda: prefill(0, x, w), reps)
            fl = 2.0 * M * N * K
            print('  M=%-6d %5dx%-5d: own %7.3f ms %7.1f TF | hipBLASLt %7.3f ms %7.1f TF | own / library %.3f' % (M, K, N, t1, fl / t1 / 1e9, t0,
                                                                                                                 fl / t0 / 1e9, t0 / t1))
        if M >= 1024:
            K, N = 4096, 11008
            g, u = rand_set(K, N), rand_set(K, N)
            x = torch.randn((M, K), device=dev, generator=gen).half()
            reps = 3 if M >= 16384 else 10
            t1 = timeit(lambda: prefill(1, x, g, up=u), reps)
            t0 = timeit(lambda: prefill(0, x, g, up=u), reps)
            fl = 4.0 * M * N * K
            print('  M=%-6d gate/up 2x%dx%d: own %7.3f ms %7.1f TF | hipBLASLt + fp32 SiLU pass %7.3f ms %7.1f TF | own / library %.3f'
                  % (M, K, N, t1, fl / t1 / 1e9, t0, fl / t0 / 1e9, t0 / t1))
            K, N = 4096, 4096
            w = rand_set(K, N)
            dy = torch.randn((M, N), device=dev, generator=gen).half()
            t1 = timeit(lambda: backward(1, dy, w), reps)
            t0 = timeit(lambda: backward(0, dy, w), reps)
            fl = 2.0 * M * N * K
            print('  M=%-6d backward %dx%d: own %7.3f ms %7.1f TF | hipBLASLt %7.3f ms %7.1f TF | own / library %.3f' % (M, K, N, t1, fl / t1 / 1e9, t0,
                                                                                                                       fl / t0 / 1e9, t0 / t1))


if __name__ == '__main__':
    for shape in (int(v) for v in os.environ.get('MFMA', '16,32').split(',')):
        prev = lib.gptq_set_gemm8_mfma(shape)
        print('######## tile GEMM on v_mfma_f32_%s_f16 ########' % ('16x16x32' if shape == 16 else '32x32x16'))
        t = time.time()
        good = check()
        print('correctness: %s (%.1f s)' % ('ALL OK' if good else 'FAILURES', time.time() - t))
        if good or os.environ.get('FORCE_PERF'):
            perf()
        lib.gptq_set_gemm8_mfma(prev)
    lib.gptq_set_prefill_route(1)
