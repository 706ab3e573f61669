#!/usr/bin/env python3
"""Sweep the GEMV tiling variants x K-split for the LLaMA-7B decode shapes on one MI355X.
Cold-HBM protocol: every shape rotates over 32 distinct weight sets (>= 280 MB) inside one
hipGraph; time = HIP events around 10 replays.  Prints one JSON line per configuration and a
summary table; the winners are what csrc/capi.hip's pick_gemv() encodes.
Usage: python tools/sweep_gemv.py [--out gpurun_out/sweep.jsonl] [--m 1]"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'gptq-for-llama_amd'))
sys.path.insert(0, ROOT)
import torch
from bench import PackedSet, alg_bytes, BITS, GS, HBM_PEAK_GBS
from quant import _native


def time_graph(fn, nsets, reps=10):
    for i in range(nsets):
        fn(i)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for i in range(nsets):
            fn(i)
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (reps * nsets)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--out', default=os.path.join(ROOT, 'gpurun_out', 'sweep.jsonl'))
    ap.add_argument('--m', type=int, default=1)
    ap.add_argument('--nsets', type=int, default=32)
    ap.add_argument('--kernel', default='gemv', choices=['gemv', 'skinny'])
    args = ap.parse_args()
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    dev = 'cuda:0'
    lib = _native.lib()
    ws = _native.workspace(torch.device(dev))
    gen = torch.Generator(device=dev)
    gen.manual_seed(0)
    M = args.m
    shapes = [('4096x4096', 4096, 4096, False), ('4096x12288', 4096, 12288, False), ('11008x4096', 11008, 4096, False),
              ('4096x11008', 4096, 11008, False), ('fused2x4096x11008', 4096, 11008, True)]
    rows = []
    with open(args.out, 'a') as fo:
        for name, K, N, fused in shapes:
            nsets = args.nsets
            sets = [PackedSet(K, N, dev, gen) for _ in range(nsets * (2 if fused else 1))]
            x = torch.randn((M, K), device=dev, generator=gen).half()
            y = torch.empty((M, N), dtype=torch.float16, device=dev)
            entry = lib.gptq_gemv_f16 if args.kernel == 'gemv' else lib.gptq_skinny_f16

            def mm(i):
                w = sets[i]
                rc = entry(x.data_ptr(), K, w.qweight.data_ptr(), w.scales.data_ptr(), w.qzeros.data_ptr(), None, None,
                           y.data_ptr(), N, M, K, N, BITS, GS, ws.data_ptr(), ws.numel(), torch.cuda.current_stream().cuda_stream)
                _native.check(rc, 'gemv')

            def mlp(i):
                g, u = sets[2 * i], sets[2 * i + 1]
                rc = lib.gptq_fused_mlp_f16(x.data_ptr(), K, g.qweight.data_ptr(), g.scales.data_ptr(), g.qzeros.data_ptr(), None,
                                            u.qweight.data_ptr(), u.scales.data_ptr(), u.qzeros.data_ptr(), None, y.data_ptr(), N,
                                            M, K, N, BITS, GS, ws.data_ptr(), ws.numel(), torch.cuda.current_stream().cuda_stream)
                _native.check(rc, 'fused')

            fn = mlp if fused else mm
            nbytes = alg_bytes(M, K, N, nsets=2 if fused else 1)
            variants = range(3) if args.kernel == 'gemv' else [4]
            for v in variants:
                for sk in ((4, 8, 16) if fused else ((1, 2, 3, 4, 6, 8, 12) if args.kernel == 'skinny' else (8, 16, 22, 32, 43, 64))):
                    lib.gptq_set_gemv_variant(v)
                    lib.gptq_set_split_k(sk)
                    try:
                        us = time_graph(fn, nsets)
                    except RuntimeError as e:
                        rec = {'shape': name, 'M': M, 'variant': v, 'split_k': sk, 'error': str(e)[:80]}
                        fo.write(json.dumps(rec) + '\n')
                        continue
                    finally:
                        lib.gptq_set_gemv_variant(-1)
                        lib.gptq_set_split_k(-1)
                    rec = {'kernel': args.kernel, 'shape': name, 'M': M, 'variant': v, 'split_k': sk, 'us': round(us, 3),
                           'GBps': round(nbytes / us / 1e3, 1), 'frac': round(nbytes / us / 1e3 / HBM_PEAK_GBS, 4)}
                    rows.append(rec)
                    fo.write(json.dumps(rec) + '\n')
                    fo.flush()
            del sets
            torch.cuda.empty_cache()
    print('%-20s %3s %7s %5s %9s %9s %6s' % ('shape', 'M', 'variant', 'split', 'us', 'GB/s', 'frac'))
    for name in dict.fromkeys(r['shape'] for r in rows):
        best = sorted((r for r in rows if r['shape'] == name), key=lambda r: r['us'])[:6]
        for r in best:
            print('%-20s %3d %7d %5d %9.3f %9.1f %6.3f' % (r['shape'], r['M'], r['variant'], r['split_k'], r['us'], r['GBps'], r['frac']))


if __name__ == '__main__':
    main()
