#!/usr/bin/env python3
"""gptq_layer_decode_f16 (y = residual + layer(rmsnorm(x)), the operator of the batched decode engine) per LLaMA-7B shape and batch:
us per launch on cold weights (hipGraph over >= 300 MB of prepared layers), plain / + norm / + residual / + both, and the LM head
(gptq_dense_matmat_f16) per batch.  MS=1,2,4,8,16  SHAPES=4096x4096,...  BITS=4"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'gptq-for-llama_amd')); sys.path.insert(0, ROOT)
import torch
from bench import alg_bytes, GS
from quant import _native
from quant.layer import prepared
dev = torch.device('cuda:0'); lib = _native.lib()
gen = torch.Generator(device=dev); gen.manual_seed(0)
MS = [int(v) for v in os.environ.get('MS', '1,2,4,5,8,16').split(',')]
BITS = int(os.environ.get('BITS', '4'))
SHAPES = [(4096, 4096, 1), (4096, 12288, 1), (4096, 11008, 2), (11008, 4096, 1)]
if os.environ.get('SHAPES'):
    SHAPES = [tuple(int(v) for v in s.split('x')) + (1,) for s in os.environ['SHAPES'].split(',')]


def rand_set(K, N):
    G = K // GS
    return (torch.randint(-2**31, 2**31 - 1, (K * BITS // 32, N), dtype=torch.int32, device=dev, generator=gen),
            (torch.rand((G, N), device=dev, generator=gen) * 0.01 + 0.001).half(),
            torch.randint(-2**31, 2**31 - 1, (G, N * BITS // 32), dtype=torch.int32, device=dev, generator=gen), None)


def timed(fn, n):
    for i in range(n): fn(i)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for i in range(n): fn(i)
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): g.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (5 * n)


for K, N, NS in SHAPES:
    n = int(300e6 // (NS * alg_bytes(1, K, N, bits=BITS))) + 1
    keep = [tuple(rand_set(K, N) for _ in range(NS)) for _ in range(n)]
    pls = [prepared(s, None, BITS, GS, K, N) for s in keep]
    for pl in pls: pl.release()
    for s in keep:
        for t in s: pass
    nw = torch.ones(K, dtype=torch.float16, device=dev)
    for M in MS:
        x = torch.randn((M, K), device=dev, generator=gen).half(); y = torch.empty((M, N), dtype=torch.float16, device=dev)
        r = torch.randn((M, N), device=dev, generator=gen).half()
        need = max(lib.gptq_layer_decode_scratch_bytes(pl.handle, M) for pl in pls[:1])
        scratch = torch.empty(max(need, 256), dtype=torch.uint8, device=dev)
        row = {'bits': BITS, 'shape': '%dx%d%s' % (K, N, ' pair' if NS == 2 else ''), 'M': M, 'route': lib.gptq_layer_route_for(pls[0].handle, M)}
        for name, norm, res in (('plain', None, None), ('norm', nw, None), ('res', None, r), ('norm_res', nw, r)):
            if res is not None and NS == 2: continue

            def run(i):
                s = _native.stream_ptr(dev)
                ws = _native.layer_workspace(dev, s)
                rc = lib.gptq_layer_decode_f16(pls[i].handle, x.data_ptr(), K, y.data_ptr(), N, M, _native.ptr(norm), 1e-6, _native.ptr(res), N if res is not None else 0,
                                               ws.data_ptr(), ws.numel(), scratch.data_ptr(), scratch.numel(), s)
                _native.check(rc, 'decode')
            row[name + '_us'] = round(timed(run, n), 2)
        print(json.dumps(row), flush=True)
    del pls, keep

# LM head
N, K = 32000, 4096
Ws = [(torch.randn((N, K), device=dev, generator=gen) * 0.02).half() for _ in range(3)]
nw = torch.ones(K, dtype=torch.float16, device=dev)
for M in MS:
    x = torch.randn((M, K), device=dev, generator=gen).half(); y = torch.empty((M, N), dtype=torch.float16, device=dev)

    def run(i):
        _native.check(lib.gptq_dense_matmat_f16(x.data_ptr(), K, Ws[i].data_ptr(), K, None, y.data_ptr(), N, M, N, K, nw.data_ptr(), 1e-6, _native.stream_ptr(dev)), 'lm')

    def run_lib(i):
        torch.matmul(x, Ws[i].t(), out=y)
    print(json.dumps({'lm_head_rows': M, 'own_norm_fused_us': round(timed(run, 3), 2), 'torch_matmul_us': round(timed(run_lib, 3), 2)}), flush=True)
