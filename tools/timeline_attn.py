#!/usr/bin/env python3
"""Per-wave s_memtime checkpoints of ONE fused decode-attention launch inside a DecodeEngine step (development aid)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'gptq-for-llama_amd')); sys.path.insert(0, ROOT)
import numpy as np, torch
from quant import _native
from quant.decode import build_random_llama, DecodeEngine
dev = 'cuda:0'; lib = _native.lib()
m = build_random_llama(dev)
eng = DecodeEngine(m, t_max=2048)          # eager launches: the debug pointer is read at launch time
tok = torch.zeros(1, dtype=torch.long, device=dev)
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 20):
    eng.decode(tok)
torch.cuda.synchronize()
dbg = torch.zeros(32 * 4 * 10, dtype=torch.int64, device=dev)
lib.gptq_set_debug_buffer(dbg.data_ptr())
eng.decode(tok)                             # every layer overwrites the stamps: the LAST layer's launch is analysed
torch.cuda.synchronize()
lib.gptq_set_debug_buffer(None)
d = dbg.cpu().numpy().reshape(-1, 10)
d = d[d[:, 0] != 0]
r0 = d[:, 0].min()
names = ['start(real,us)', 'pos loaded', 'rope + barrier', 'scores + barrier', 'softmax done', 'out stored', 'end(real,us)']
cols = [(d[:, 0] - r0) / 100.0, d[:, 2] - d[:, 1], d[:, 3] - d[:, 1], d[:, 4] - d[:, 1], d[:, 5] - d[:, 1], d[:, 6] - d[:, 1], (d[:, 8] - r0) / 100.0]
print('waves', len(d), 'kernel span %.2f us' % ((d[:, 8].max() - r0) / 100.0))
names += ['  q in regs', '  scores computed', '  pv accumulated']
cols += [d[:, 7], d[:, 9] >> 32, d[:, 9] & 0xffffffff]
for n, v in zip(names, cols):
    v = np.sort(v)
    print('  %-18s min %8.2f  p50 %8.2f  max %8.2f %s' % (n, v[0], v[len(v) // 2], v[-1], 'us' if 'real' in n else 'cycles'))
