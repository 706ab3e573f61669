#!/usr/bin/env python3
"""Per-wave s_memtime checkpoints of ONE launch of the stream kernel (development aid).
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
        rc = lib.gptq_skinny_f16(x.data_ptr(), a.K, g.qweight.data_ptr(), g.scales.data_ptr(), g.qzeros.data_ptr(), None, None,
                                 y.data_ptr(), a.N, a.M, a.K, a.N, BITS, GS, ws.data_ptr(), ws.numel(), s)
    _native.check(rc, 'launch')
for i in range(nsets): launch(i)
torch.cuda.synchronize()
dbg = torch.zeros(8192 * 8 * 8, dtype=torch.int64, device=dev)
lib.gptq_set_debug_buffer(dbg.data_ptr())
launch(3)
torch.cuda.synchronize()
lib.gptq_set_debug_buffer(None)
d = dbg.cpu().numpy().reshape(-1, 8)
d = d[d[:, 0] != 0]
names = ['start', 'loads issued', 'x staged', 'stage0 done', 'compute done', 'reduced', 'end']
print('waves recorded', len(d), '(s_memtime ticks, ~1 ns; per-wave times relative to the wave\'s own start)')
for i, n in enumerate(names[1:], 1):
    ok = d[:, i] != 0
    col = (d[ok, i] - d[ok, 0]).astype(np.float64)
    if len(col): print('%-14s min %7.0f  p50 %7.0f  mean %8.1f  p95 %7.0f  max %7.0f' % (n, col.min(), np.median(col), col.mean(), np.percentile(col, 95), col.max()))
# launch skew: start time relative to the earliest wave on a counter base that looks shared (cluster by magnitude)
st = d[:, 0].astype(np.float64)
for lo in sorted(set((st // 1e9).tolist())):
    grp = st[(st // 1e9) == lo]
    print('start skew (cluster %d, %d waves): p50 %.0f  p95 %.0f  max %.0f' % (lo, len(grp), np.median(grp - grp.min()), np.percentile(grp - grp.min(), 95), (grp - grp.min()).max()))
    en = d[(st // 1e9) == lo][:, 6].astype(np.float64)
    print('   last end - first start: %.0f' % (en.max() - grp.min()))
