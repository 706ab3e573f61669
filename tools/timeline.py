#!/usr/bin/env python3
"""Per-wave s_memtime checkpoints of ONE launch of the rowwave GEMV (development aid).
python tools/timeline.py --K 4096 --N 4096 [--split S] [--variant W] [--fused]"""
import argparse, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'gptq-for-llama_amd')); sys.path.insert(0, ROOT)
import numpy as np, torch
from bench import PackedSet, BITS, GS
from quant import _native
ap = argparse.ArgumentParser()
ap.add_argument('--K', type=int, default=4096); ap.add_argument('--N', type=int, default=4096)
ap.add_argument('--M', type=int, default=1); ap.add_argument('--variant', type=int, default=-1)
ap.add_argument('--split', type=int, default=-1); ap.add_argument('--fused', action='store_true')
a = ap.parse_args()
dev = 'cuda:0'; lib = _native.lib(); ws = _native.workspace(torch.device(dev))
gen = torch.Generator(device=dev); gen.manual_seed(0)
nsets = 16
sets = [PackedSet(a.K, a.N, dev, gen) for _ in range(nsets * 2)]
x = torch.randn((a.M, a.K), device=dev, generator=gen).half(); y = torch.empty((a.M, a.N), dtype=torch.float16, device=dev)
lib.gptq_set_gemv_variant(a.variant); lib.gptq_set_split_k(a.split)
s = torch.cuda.current_stream().cuda_stream
def launch(i):
    g, u = sets[2 * i], sets[2 * i + 1]
    if a.fused:
        rc = lib.gptq_fused_mlp_f16(x.data_ptr(), a.K, g.qweight.data_ptr(), g.scales.data_ptr(), g.qzeros.data_ptr(), None,
                                    u.qweight.data_ptr(), u.scales.data_ptr(), u.qzeros.data_ptr(), None, y.data_ptr(), a.N,
                                    a.M, a.K, a.N, BITS, GS, ws.data_ptr(), ws.numel(), s)
    else:
        rc = lib.gptq_gemv_f16(x.data_ptr(), a.K, g.qweight.data_ptr(), g.scales.data_ptr(), g.qzeros.data_ptr(), None, None,
                                 y.data_ptr(), a.N, a.M, a.K, a.N, BITS, GS, ws.data_ptr(), ws.numel(), s)
    _native.check(rc, 'launch')
for i in range(nsets): launch(i)
torch.cuda.synchronize()
dbg = torch.zeros(8192 * 4 * 10, dtype=torch.int64, device=dev)
for i in range(4): launch(i)          # the stamps of the LAST of a few back-to-back cold launches are analysed
lib.gptq_set_debug_buffer(dbg.data_ptr())
launch(4)
torch.cuda.synchronize()
lib.gptq_set_debug_buffer(None)
d = dbg.cpu().numpy().reshape(-1, 10)
d = d[d[:, 1] != 0]
r0, r1 = d[:, 0].min(), d[:, 8].max()
print('waves recorded %d; kernel span (first wave start -> last wave end, s_memrealtime @100 MHz): %.2f us' % (len(d), (r1 - r0) / 100.0))
names = ['start (us, realtime)', 'loads issued', 'x arrived', 'first weights', 'last weights', 'math done', 'combine done', 'end (us, realtime)']
for k, n in enumerate(names):
    if k == 0: col = (d[:, 0] - r0) / 100.0
    elif k == 7: col = (d[:, 8] - r0) / 100.0
    else: col = (d[:, k + 1] - d[:, 1]).astype(np.float64)     # shader cycles since the wave's own start
    print('%-22s min %8.2f  p10 %8.2f  p50 %8.2f  p90 %8.2f  max %8.2f  %s' % (n, col.min(), np.percentile(col, 10), np.median(col),
          np.percentile(col, 90), col.max(), 'us' if k in (0, 7) else 'cycles'))
