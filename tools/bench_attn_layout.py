"""Does the K / V cache LAYOUT bound the decode attention?  The engine's cache is [t][heads * 128] (a head's row = 256 contiguous bytes every 8 KB);
the same launch with heads = 1 and batch = 32 reads 32 fully contiguous [t][128] streams -- the head-major layout -- with the same bytes, the same
grid and the same kernel.  Cold caches: NBUF rotating K / V buffers (> the 256 MiB Infinity Cache), hipGraph of NBUF launches.
   python tools/bench_attn_layout.py [pos ...]"""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'gptq-for-llama_amd')]
import numpy as np, torch
from quant import _native
lib = _native.lib()
DEV = torch.device('cuda:0')
t_max, hd, NBUF = 2048, 128, 24
scale = 1 / np.sqrt(hd)
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    sp = _native.stream_ptr(DEV)
    tab = torch.empty((t_max, hd // 2, 2), dtype=torch.float32, device=DEV)
    lib.gptq_rope_table_f32(tab.data_ptr(), t_max, hd, 10000.0, sp)
    for pos in [int(a) for a in sys.argv[1:]] or [0, 127, 511, 1023, 2046]:
        row = {'pos': pos}
        for name, heads, batch in (('interleaved [t][32 x 128]', 32, 1), ('head-major 32 x [t][128]', 1, 32)):
            H = heads * hd
            qkv = torch.randn((batch, 3 * H), device=DEV).half()
            kc = [(torch.randn((batch, t_max, H), device=DEV) * 0.5).half() for _ in range(NBUF)]
            vc = [(torch.randn((batch, t_max, H), device=DEV) * 0.5).half() for _ in range(NBUF)]
            p = torch.full((batch,), pos, dtype=torch.int64, device=DEV)
            out = torch.empty((batch, H), dtype=torch.float16, device=DEV)
            nb = lib.gptq_decode_attn_batch_workspace_bytes(batch, heads, hd, t_max)
            ws = torch.zeros(nb, dtype=torch.uint8, device=DEV)
            for rec in (0, 1):
                def launch(i):
                    if rec:
                        rc = lib.gptq_decode_attn_split_f16(qkv.data_ptr(), 3 * H, p.data_ptr(), kc[i].data_ptr(), vc[i].data_ptr(), ws.data_ptr(), nb, batch, heads, hd, t_max,
                                                            10000.0, scale, tab.data_ptr(), 0, sp)
                    else:
                        rc = lib.gptq_decode_attn_batch_f16(qkv.data_ptr(), 3 * H, p.data_ptr(), kc[i].data_ptr(), vc[i].data_ptr(), out.data_ptr(), H, ws.data_ptr(), nb, batch,
                                                            heads, hd, t_max, 10000.0, scale, tab.data_ptr(), None, sp)
                    assert rc == 0, rc
                for i in range(NBUF):
                    launch(i)
                torch.cuda.synchronize()
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, stream=s):
                    for i in range(NBUF):
                        launch(i)
                g.replay(); torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(s)
                for _ in range(20):
                    g.replay()
                e1.record(s); torch.cuda.synchronize()
                us = e0.elapsed_time(e1) * 1000 / (20 * NBUF)
                row['%s %s' % (name, 'records' if rec else 'self-merging')] = round(us, 2)
            del kc, vc
        row['MB'] = round(32 * (pos + 1) * 512 / 1e6, 2)
        print(json.dumps(row), flush=True)
