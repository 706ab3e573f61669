#!/usr/bin/env python3
"""round 5 debugging aid: DecodeEngine(batch=B) against the module chain on the tiny test model, error per step and per row, with the
engine's pieces switched (fused norm, LM head kernel)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'gptq-for-llama_amd')); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, torch
from quant import decode as D
from test_gpu_batch import run_steps, HD128
DEV = 'cuda:0'
for B in (8, 16):
    q = D.build_random_llama(DEV, seed=3 + B, **HD128)
    ids = torch.randint(0, HD128['vocab_size'], (B, 9), device=DEV, generator=torch.Generator(device=DEV).manual_seed(B))
    expect = run_steps(q, ids, 1)
    for fuse in (True, False):
        for lm in (True, False):
            D.LM_HEAD_KERNEL = lm
            eng = D.DecodeEngine(q, t_max=64, batch=B, fuse_norm=fuse)
            got = np.stack([eng.decode(ids[:, i]).float().cpu().numpy() for i in range(ids.shape[1])])
            err = np.abs(got - expect).max(axis=2) / np.abs(expect).max()
            print('B', B, 'fuse_norm', fuse, 'lm_kernel', lm, 'max %.2e' % err.max(), 'per step', ' '.join('%.1e' % e for e in err.max(axis=1)),
                  'worst row', int(err.max(axis=0).argmax()), flush=True)
    if B == 16:       # who is closer to a dense fp16 twin (stock HF modules, oracle-dequantised weights) at the worst (step, row)?
        from test_gpu_model import dense_twin
        qu = D.build_random_llama(DEV, seed=3 + B, fused=False, **HD128)
        ref = run_steps(dense_twin(qu, HD128), ids, 1)
        D.LM_HEAD_KERNEL = True
        eng = D.DecodeEngine(q, t_max=64, batch=B)
        got = np.stack([eng.decode(ids[:, i]).float().cpu().numpy() for i in range(ids.shape[1])])
        sc = np.abs(ref).max()
        for name, a in (('engine', got), ('chain', expect)):
            e = np.abs(a - ref).max(axis=2) / sc
            print('%s vs dense twin: max %.2e at (step, row) %s; at (4, 13): %.2e; median %.2e' % (name, e.max(), np.unravel_index(e.argmax(), e.shape), e[4, 13], np.median(e)), flush=True)
    # the eager chain against itself at another batch composition: rows 0..7 alone (M = 8 kernels) vs inside the batch of 16
    if B == 16:
        e8 = run_steps(q, ids[:8], 1)
        d = np.abs(e8 - expect[:, :8]).max() / np.abs(expect).max()
        print('module chain: rows 0..7 run as a batch of 8 vs inside the batch of 16: %.2e' % d, flush=True)
        eng8 = D.DecodeEngine(q, t_max=64, batch=8)
        g8 = np.stack([eng8.decode(ids[:8, i]).float().cpu().numpy() for i in range(ids.shape[1])])
        print('engine(8) on rows 0..7 vs chain(16) rows 0..7: %.2e;  vs chain(8): %.2e' % (np.abs(g8 - expect[:, :8]).max() / np.abs(expect).max(),
                                                                                         np.abs(g8 - e8).max() / np.abs(expect).max()), flush=True)
