import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'gptq-for-llama_amd'), os.path.join(ROOT, 'tests')]
from quant import _native
from quant.layer import prepared
from util import make_random_layer
DEV = 'cuda:0'
lib = _native.lib()
heads, hd, t_max, N, bits, gs = 4, 128, 2048, 256, 4, 128
K = heads * hd
L = make_random_layer(bits, gs, K, N, seed=904)
dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(DEV)
sets = ((dev(L['qweight']), dev(L['scales']), dev(L['qzeros']), dev(L['g_idx'])),)
pl = prepared(sets, None, bits, gs, K, N)
s = _native.stream_ptr(torch.device(DEV))
g = torch.Generator(device=DEV).manual_seed(1)
qkv = torch.randn((1, 3 * K), device=DEV, generator=g).half()
kc = (torch.randn((1, t_max, K), device=DEV, generator=g) * 0.5).half()
vc = (torch.randn((1, t_max, K), device=DEV, generator=g) * 0.5).half()
tab = torch.empty((t_max, hd // 2, 2), dtype=torch.float32, device=DEV)
lib.gptq_rope_table_f32(tab.data_ptr(), t_max, hd, 10000.0, s)
scale = 1 / np.sqrt(hd)
nb = lib.gptq_decode_attn_batch_workspace_bytes(1, heads, hd, t_max)
S = lib.gptq_decode_attn_splits(1, heads, hd, t_max)
lws = _native.layer_workspace(torch.device(DEV), s)
scratch = torch.empty(1 << 20, dtype=torch.uint8, device=DEV)
for pos in (0, 300, 1000):
    p = torch.tensor([pos], dtype=torch.int64, device=DEV)
    wsa = torch.zeros(nb, dtype=torch.uint8, device=DEV)
    xa = torch.zeros((1, K), dtype=torch.float16, device=DEV)
    ka, va = kc.clone(), vc.clone()
    print('attn', lib.gptq_decode_attn_batch_f16(qkv.data_ptr(), 3 * K, p.data_ptr(), ka.data_ptr(), va.data_ptr(), xa.data_ptr(), K, wsa.data_ptr(), nb, 1, heads, hd, t_max, 10000.0, scale, tab.data_ptr(), None, s))
    ya = torch.zeros((1, N), dtype=torch.float16, device=DEV)
    print('dec', lib.gptq_layer_decode_f16(pl.handle, xa.data_ptr(), K, ya.data_ptr(), N, 1, None, 0.0, None, 0, lws.data_ptr(), lws.numel(), scratch.data_ptr(), scratch.numel(), s))
    wsb = torch.zeros(nb // 4, dtype=torch.float32, device=DEV)
    kb, vb = kc.clone(), vc.clone()
    print('split', lib.gptq_decode_attn_split_f16(qkv.data_ptr(), 3 * K, p.data_ptr(), kb.data_ptr(), vb.data_ptr(), wsb.data_ptr(), nb, 1, heads, hd, t_max, 10000.0, scale, tab.data_ptr(), 768, s))
    yb = torch.zeros((1, N), dtype=torch.float16, device=DEV)
    print('decattn', lib.gptq_layer_decode_attn_f16(pl.handle, wsb.data_ptr(), nb, p.data_ptr(), 1, heads, hd, t_max, 768, yb.data_ptr(), N, None, 0, s))
    torch.cuda.synchronize()
    num = wsb[:S * K].view(S, K).double().cpu().numpy(); md = wsb[S * K:S * K + S * heads * 2].view(S, heads, 2).double().cpu().numpy()
    nsp = 1 if pos < 768 else 2
    Mx = md[:nsp, :, 0].max(0); w = np.exp2(md[:nsp, :, 0] - Mx[None]); den = (w * md[:nsp, :, 1]).sum(0)
    xh = ((w[:, :, None] * num[:nsp].reshape(nsp, heads, hd)).sum(0) / den[:, None]).reshape(-1)
    print(pos, 'xa vs host merge', np.abs(xa[0].double().cpu().numpy() - xh).max(), 'ya vs yb', (ya.float() - yb.float()).abs().max().item(), ya[0, :4].tolist(), yb[0, :4].tolist())
    xh16 = torch.from_numpy(xh.astype(np.float16)).to(DEV).view(1, K)
    yc = torch.zeros((1, N), dtype=torch.float16, device=DEV)
    lib.gptq_layer_decode_f16(pl.handle, xh16.data_ptr(), K, yc.data_ptr(), N, 1, None, 0.0, None, 0, lws.data_ptr(), lws.numel(), scratch.data_ptr(), scratch.numel(), s)
    torch.cuda.synchronize()
    print('   yc(host x) vs ya', (yc.float() - ya.float()).abs().max().item(), 'vs yb', (yc.float() - yb.float()).abs().max().item())
    # which x reproduces yb?  solve nothing: try x = num[0] / den directly etc.
