#!/usr/bin/env python3
"""BASELINE config 3: LLaMA-7B-shaped 4-bit g128 batched matmul, M = batch*seq (default 32*2048 =
65536), through gptq_matmul248_f16 (MFMA tile kernel).  Reports TFLOP/s = 2*M*N*K / t against the
2.5 PFLOP/s dense fp16 MFMA peak, next to torch.matmul (hipBLASLt) on the dequantised weight.
usage: python tools/bench_prefill.py [--m 65536] [--reps 5]"""
import argparse, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'gptq-for-llama_amd')); sys.path.insert(0, ROOT)
import torch
from bench import PackedSet, BITS, GS
from quant import _native, quant_linear as QL

ap = argparse.ArgumentParser()
ap.add_argument('--m', type=int, default=65536); ap.add_argument('--reps', type=int, default=5)
ap.add_argument('--no-dense', action='store_true')
ap.add_argument('--kernel', type=int, default=0, help='gptq_set_gemm_kernel: 2 ping-pong (default), 3 packed-B in LDS')
a = ap.parse_args()
if a.kernel:
    _native.lib().gptq_set_gemm_kernel(a.kernel)
dev = 'cuda:0'
gen = torch.Generator(device=dev); gen.manual_seed(0)
PEAK = 2500.0
out = []
for K, N in [(4096, 4096), (4096, 12288), (4096, 11008), (11008, 4096)]:
    w = PackedSet(K, N, dev, gen)
    x = torch.randn((a.m, K), device=dev, generator=gen).half()
    g_idx = (torch.arange(K, device=dev) // GS).to(torch.int32)
    f = lambda: QL.matmul248(x, w.qweight, w.scales, w.qzeros, g_idx, BITS, 15, family='abi')     # the fused tile kernel, whatever GPTQ_PREFILL says
    y = f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(a.reps): y = f()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / a.reps
    tf = 2.0 * a.m * N * K / ms / 1e9
    rec = {'kernel': a.kernel or 2, 'shape': '%dx%d' % (K, N), 'M': a.m, 'ms': round(ms, 3), 'TFLOPs': round(tf, 1), 'frac_of_2.5PF': round(tf / PEAK, 4)}
    if not a.no_dense:
        # dense ceiling: the same product with the weight dequantised once (fp16) through hipBLASLt
        eye = torch.eye(K, device=dev, dtype=torch.float16)
        W = torch.cat([QL.matmul248(eye[i:i + 1024], w.qweight, w.scales, w.qzeros, g_idx, BITS, 15) for i in range(0, K, 1024)])
        del eye
        yd = x @ W; torch.cuda.synchronize()
        e0.record()
        for _ in range(a.reps): yd = x @ W
        e1.record(); torch.cuda.synchronize()
        msd = e0.elapsed_time(e1) / a.reps
        rec['dense_fp16_matmul_ms'] = round(msd, 3)
        rec['dense_TFLOPs'] = round(2.0 * a.m * N * K / msd / 1e9, 1)
        rec['max_abs_diff_vs_dense'] = float((y.float() - yd.float()).abs().max())
        del W, yd
    print(json.dumps(rec)); out.append(rec)
    del x, y, w
    torch.cuda.empty_cache()

# fused gate/up + SiLU at prefill size (quant.fused_mlp.fused_gate_up -> gptq_fused_mlp_f16: two tile GEMMs, in-place epilogue)
from quant import fused_mlp as FM
K, N = 4096, 11008
wg, wu = PackedSet(K, N, dev, gen), PackedSet(K, N, dev, gen)
x = torch.randn((a.m, K), device=dev, generator=gen).half()
gi = (torch.arange(K, device=dev) // GS).to(torch.int32)
f = lambda: FM.fused_gate_up(x, (wg.qweight, wg.scales, wg.qzeros, gi), (wu.qweight, wu.scales, wu.qzeros, gi), BITS, GS, family='abi')
y = f(); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(a.reps): y = f()
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / a.reps
print(json.dumps({'fused_gate_up_silu': '2x%dx%d' % (K, N), 'M': a.m, 'ms': round(ms, 3), 'TFLOPs': round(4.0 * a.m * N * K / ms / 1e9, 1)}))
