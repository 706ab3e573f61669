#!/usr/bin/env python3
"""17 .. 128 rows on the LLaMA-7B shapes through the drop-in matmul248 (hipGraph, cold weights) -- run once per GPTQ_MMR setting (the loader /
consumer small-batch kernel, csrc/stripe_mm.inc stripe_mmr_kernel).   MS=16,32,48,64,128  python tools/bench_mmr.py"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'gptq-for-llama_amd')); sys.path.insert(0, ROOT)
import torch
from bench import PackedSet, alg_bytes, BITS, GS, HIDDEN, INTER, _time_cold
from quant import quant_linear as QL
dev = 'cuda:0'
gen = torch.Generator(device=dev); gen.manual_seed(6)
MS = [int(v) for v in os.environ.get('MS', '16,32,48,64,128').split(',')]
XPAD = int(os.environ.get('XPAD', '0'))
out = {'GPTQ_MMR': os.environ.get('GPTQ_MMR', ''), 'XPAD': XPAD}
SHAPES = [(HIDDEN, 3 * HIDDEN), (HIDDEN, INTER)] if os.environ.get('SHAPES2') else [(HIDDEN, 3 * HIDDEN), (HIDDEN, INTER), (HIDDEN, HIDDEN), (INTER, HIDDEN)]
if os.environ.get('SHAPES'):           # SHAPES=4096x8192,8192x8192
    SHAPES = [tuple(int(v) for v in sh.split('x')) for sh in os.environ['SHAPES'].split(',')]
for K, N in SHAPES:
    nsets = int(300e6 // alg_bytes(1, K, N)) + 1
    sets = [PackedSet(K, N, dev, gen) for _ in range(nsets)]
    gi = (torch.arange(K, device=dev) // GS).to(torch.int32)
    row = {}
    for M in MS:
        x = torch.randn((M, K + XPAD), device=dev, generator=gen).half()[:, :K]        # XPAD: elements of padding per row (the row stride's L2 channel pattern)
        def run(i):
            w = sets[i]
            QL.matmul248(x, w.qweight, w.scales, w.qzeros, gi, BITS, 15)
        row['M%d' % M] = round(_time_cold(run, nsets, reps=3), 2)
    out['%dx%d' % (K, N)] = row
    del sets
print(json.dumps(out), flush=True)
