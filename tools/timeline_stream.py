#!/usr/bin/env python3
"""per-wave stamps of the MFMA stream kernel (2 <= M <= 64): start, loads issued, x staged, first stage
done, compute done, reduced, end -- relative to each wave's own start (shader cycles)."""
import argparse, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'gptq-for-llama_amd')); sys.path.insert(0, ROOT)
import numpy as np, torch
from bench import PackedSet, BITS, GS
from quant import _native
ap = argparse.ArgumentParser(); ap.add_argument('--K', type=int, default=4096); ap.add_argument('--N', type=int, default=4096)
ap.add_argument('--M', type=int, default=16); a = ap.parse_args()
dev = 'cuda:0'; lib = _native.lib(); ws = _native.workspace(torch.device(dev))
gen = torch.Generator(device=dev); gen.manual_seed(0)
sets = [PackedSet(a.K, a.N, dev, gen) for _ in range(8)]
x = torch.randn((a.M, a.K), device=dev, generator=gen).half(); y = torch.empty((a.M, a.N), dtype=torch.float16, device=dev)
def launch(i):
    w = sets[i]
    _native.check(lib.gptq_skinny_f16(x.data_ptr(), a.K, w.qweight.data_ptr(), w.scales.data_ptr(), w.qzeros.data_ptr(), None, None, y.data_ptr(),
                                      a.N, a.M, a.K, a.N, BITS, GS, ws.data_ptr(), ws.numel(), torch.cuda.current_stream().cuda_stream), 'skinny')
for i in range(6): launch(i)
torch.cuda.synchronize()
dbg = torch.zeros(16384 * 8 * 8, dtype=torch.int64, device=dev)
lib.gptq_set_debug_buffer(dbg.data_ptr()); launch(7); torch.cuda.synchronize(); lib.gptq_set_debug_buffer(None)
d = dbg.cpu().numpy().reshape(-1, 8); d = d[d[:, 0] != 0]
print('waves', len(d))
names = ['loads issued', 'x staged', 'stage0 done', 'compute done', 'reduced', 'end']
for i, n in enumerate(names, 1):
    ok = d[:, i] != 0; col = (d[ok, i] - d[ok, 0]).astype(np.float64)
    if len(col): print('%-14s p10 %7.0f p50 %7.0f p90 %7.0f max %7.0f' % (n, np.percentile(col, 10), np.median(col), np.percentile(col, 90), col.max()))
